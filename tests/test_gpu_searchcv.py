"""BayesSearchCV on the device-backed Optimizer: the two tests of the reference (tests/test_searchcv.py), same
estimator, search space, iteration count and acceptance threshold."""
import pytest
from sklearn.datasets import load_iris
from sklearn.model_selection import train_test_split
from sklearn.svm import SVC

pytestmark = pytest.mark.gpu


def _spaces():
    from bayes_skopt_amd.space import Categorical, Integer, Real

    return {
        "C": Real(1e-6, 1e6, prior="log-uniform"),
        "gamma": Real(1e-6, 1e1, prior="log-uniform"),
        "degree": Integer(1, 8),
        "kernel": Categorical(["linear", "poly", "rbf"]),
    }


@pytest.mark.parametrize("policy", ["best_setting", "best_mean"])
def test_searchcv_run(policy):
    import bayes_skopt_amd as bask

    X, y = load_iris(return_X_y=True)
    X_train, X_test, y_train, y_test = train_test_split(X, y, train_size=0.75, random_state=0)
    opt = bask.BayesSearchCV(SVC(), _spaces(), n_iter=11, cv=None, return_policy=policy, random_state=0)
    opt.fit(X_train, y_train)
    assert opt.score(X_test, y_test) > 0.89
    assert len(opt.cv_results_["params"]) == 11
    # the 11th point came from the GP: the surrogate was fitted on the device
    gp = opt.optimizer_results_[0].models[-1]
    assert gp.chain_ is not None and gp.X_train_.shape == (11, 4) or gp.X_train_.shape[0] == 11


def test_searchcv_longer_run_uses_the_surrogate():
    import bayes_skopt_amd as bask

    X, y = load_iris(return_X_y=True)
    X_train, X_test, y_train, y_test = train_test_split(X, y, train_size=0.75, random_state=0)
    opt = bask.BayesSearchCV(SVC(), _spaces(), n_iter=16, cv=3, random_state=1,
                             optimizer_kwargs=dict(n_initial_points=8, gp_samples=60, gp_burnin=5, n_points=300))
    opt.fit(X_train, y_train)
    assert opt.score(X_test, y_test) > 0.89
    assert opt.optimizer_kwargs_["n_initial_points"] == 8 and opt.gp_samples_ == 60
