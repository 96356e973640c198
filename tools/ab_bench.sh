#!/bin/bash
# A/B of library builds / environment switches on ONE box, interleaved (boxes differ by several percent):
#   bash tools/ab_bench.sh "NAME1:ENV=..;ENV=..;SEG_LIB=ab/some.so" "NAME2:..." ...
# SEG_LIB=<path> selects another build of libseg_hip.so for that arm (read by _ffi.py; the in-tree file is never touched).
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=$(echo ${spec#*:} | tr ';' ' ')
    env $envs python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10 ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  done
done
