#!/bin/bash
# run a command against alternative builds of libseg_hip.so: bash tools/lib_run.sh "<command>" lib1.so lib2.so ...
P=tf-1d-2d-segmentation-end2endpipelines_amd
cp $P/libseg_hip.so /tmp/cur.so
CMD=$1; shift
for lib in "$@"; do
  cp $lib $P/libseg_hip.so
  echo "== $lib"; eval "$CMD"
done
cp /tmp/cur.so $P/libseg_hip.so
