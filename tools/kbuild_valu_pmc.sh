#!/bin/bash
# VALU / VMEM issue counters of the Gram build (kbuild2_kernel) at config C: is it bound by the fp64 VALU?
# usage (repo root on the box): bash tools/kbuild_valu_pmc.sh r03_kbuild
set -u
TAG=${1:-rXX_kbuild}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BGP_STREAMS=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT -o ${TAG}_valu -- python3 $ROOT/tools/pmc_probe.py > $OUT/${TAG}_valu.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $OUT -o ${TAG}_vmem -- python3 $ROOT/tools/pmc_probe.py > $OUT/${TAG}_vmem.log 2>&1
python3 - <<PY > $OUT/${TAG}_counters.txt
import sqlite3, glob
for tag in ("valu", "vmem"):
    dbs = glob.glob("$OUT/**/${TAG}_%s_results.db" % tag, recursive=True)
    if not dbs:
        print("# no database for pass", tag); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    print("# pass", tag)
    for name, cn, cnt, sm in sorted(rows):
        print("%-44s %-26s dispatches %5d  avg/dispatch %16.1f" % (name.split("(")[0][-44:], cn, cnt, sm / cnt))
PY
rm -f $OUT/${TAG}_*_results.db
cat $OUT/${TAG}_counters.txt
