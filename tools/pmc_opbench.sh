#!/bin/bash
# PMC counters of single-layer launches (tools/op_bench.py) - run through gpurun from the repo root:
#   bash tools/pmc_opbench.sh TAG "<op_bench args>" "<counters>"
set -u
TAG=$1; ARGS=$2; CTR=${3:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
rocprofv3 --pmc $CTR -d $O/raw -o run --output-format csv -- python3 $R/tools/op_bench.py $ARGS > $O/run.log 2>&1
cd $R
python3 tools/profile_summary.py pmc $O/raw $O/pmc.csv
rm -rf $O/raw
grep -i "wgrad\|fwd" $O/pmc.csv | head -20
