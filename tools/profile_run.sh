#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_run.sh r03 [WORKLOAD=C3] [DTYPE=f32] [full|lite]
# -> profiles/<tag>_<workload>_<dtype>_{kernel_stats,kernel_stats_single_stream}.csv, _hbm_traffic.json, _pmc_sq{1,2}.csv
# kernel stats and the PMC passes are separate runs (never --pmc together with the API / memory-copy trace domains).
set -u
TAG=${1:-r03}
WL=${2:-C3}
DT=${3:-f32}
MODE=${4:-full}
wl=$(echo $WL | tr 'A-Z' 'a-z')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${TAG}_${wl}_${DT}
P=profiles/${TAG}_${wl}_${DT}
mkdir -p $O
B="python3 $R/bench.py --workload $WL --dtype $DT --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats.log 2>&1
# the same with every kernel alone on one stream (no weight-gradient side stream): the per-kernel durations bench.py's
# roofline object is computed from (its profiled step also runs single-stream)
export SEG_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats -d $O/stats1 -o run --output-format csv -- $B > $O/stats1.log 2>&1
unset SEG_WGRAD_STREAM
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- $B > $O/write.log 2>&1
if [ "$MODE" = "full" ]; then
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq1 -o run --output-format csv -- $B > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq2 -o run --output-format csv -- $B > $O/sq2.log 2>&1
fi
cd $R
python3 tools/profile_summary.py stats $O/stats ${P}_kernel_stats.csv
python3 tools/profile_summary.py stats $O/stats1 ${P}_kernel_stats_single_stream.csv
python3 tools/profile_summary.py hbm $O/fetch $O/write ${P}_hbm_traffic.json 8      # 2 warm-up + 3 timed + 3 profiled steps
if [ "$MODE" = "full" ]; then
python3 tools/profile_summary.py pmc $O/sq1 ${P}_pmc_sq1.csv
python3 tools/profile_summary.py pmc $O/sq2 ${P}_pmc_sq2.csv
fi
cp ${P}_* $O/
# keep only the summaries (the raw traces are tens of MB)
rm -rf $O/stats $O/stats1 $O/fetch $O/write $O/sq1 $O/sq2
ls -la $O
