#!/bin/bash
# run a pytest selection against alternative builds of libseg_hip.so (bisecting): bash tools/lib_test.sh "<pytest args>" lib1.so lib2.so ...
P=tf-1d-2d-segmentation-end2endpipelines_amd
cp $P/libseg_hip.so /tmp/cur.so
ARGS=$1; shift
for lib in "$@"; do
  cp $lib $P/libseg_hip.so
  echo "== $lib: $(timeout 900 python -m pytest $ARGS 2>&1 | tail -1)"
done
cp /tmp/cur.so $P/libseg_hip.so
