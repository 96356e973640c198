#!/bin/bash
# A/B of library builds / environment switches on ONE box, interleaved (boxes differ by several percent):
#   bash tools/ab_bench.sh "NAME1:ENV=..;ENV=..;LIB=ab/some.so" "NAME2:..." ...
# LIB=<path> swaps in another build of libseg_hip.so for that arm (the working tree's build is put back at the end).
P=tf-1d-2d-segmentation-end2endpipelines_amd
cp $P/libseg_hip.so /tmp/new.so
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=$(echo ${spec#*:} | tr ';' ' ')
    lib=/tmp/new.so
    for e in $envs; do case $e in LIB=*) lib=${e#LIB=};; esac; done
    cp $lib $P/libseg_hip.so
    env $envs python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  done
done
cp /tmp/new.so $P/libseg_hip.so
