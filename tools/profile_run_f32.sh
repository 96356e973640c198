#!/bin/bash
# The parity mode's (dtype f32) rocprofv3 evidence: kernel statistics with every kernel alone on one stream, and the HBM
# traffic passes (run through gpurun from the repo root):   bash tools/profile_run_f32.sh r02c
set -u
TAG=${1:-r02c}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${TAG}_f32
mkdir -p $O
B="python3 $R/bench.py --dtype f32 --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
export SEG_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats -d $O/stats1 -o run --output-format csv -- $B > $O/stats1.log 2>&1
unset SEG_WGRAD_STREAM
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- $B > $O/write.log 2>&1
cd $R
python3 tools/profile_summary.py stats $O/stats1 profiles/${TAG}_c3_f32_kernel_stats_single_stream.csv
python3 tools/profile_summary.py hbm $O/fetch $O/write profiles/${TAG}_c3_f32_hbm_traffic.json
cp profiles/${TAG}_c3_f32_kernel_stats_single_stream.csv profiles/${TAG}_c3_f32_hbm_traffic.json $O/
rm -rf $O/stats1 $O/fetch $O/write
ls -la $O
