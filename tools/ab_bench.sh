#!/bin/bash
# A/B of library builds / environment switches on ONE box, interleaved (boxes differ by several percent):
#   bash tools/ab_bench.sh "NAME1:ENV=..;ENV=.." "NAME2:..." ...      (NAME old: uses ab/libseg_hip_old.so if present)
P=tf-1d-2d-segmentation-end2endpipelines_amd
cp $P/libseg_hip.so /tmp/new.so
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    if [ "$name" = old ]; then cp ab/libseg_hip_old.so $P/libseg_hip.so; else cp /tmp/new.so $P/libseg_hip.so; fi
    env $(echo $envs | tr ';' ' ') python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  done
done
cp /tmp/new.so $P/libseg_hip.so
