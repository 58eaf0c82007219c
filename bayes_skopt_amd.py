"""Import shim: the package directory is named ``bayes-skopt_amd`` (not a valid Python identifier);
this module loads it under the importable name ``bayes_skopt_amd``.

    import bayes_skopt_amd as bask      # BayesGPR, Optimizer, acquisition, ...
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayes-skopt_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
