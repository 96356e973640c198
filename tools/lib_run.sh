#!/bin/bash
# run a command against alternative builds of libseg_hip.so: bash tools/lib_run.sh "<command>" lib1.so lib2.so ...
# (the build is selected through SEG_LIB, read by _ffi.py; the in-tree file is never touched)
CMD=$1; shift
for lib in "$@"; do
  echo "== $lib"; SEG_LIB=$(realpath $lib) bash -c "$CMD"
done
