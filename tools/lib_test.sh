#!/bin/bash
# run a pytest selection against alternative builds of libseg_hip.so (bisecting): bash tools/lib_test.sh "<pytest args>" lib1.so lib2.so ...
# (the build is selected through SEG_LIB, read by _ffi.py; the in-tree file is never touched)
ARGS=$1; shift
for lib in "$@"; do
  echo "== $lib: $(SEG_LIB=$(realpath $lib) timeout 900 python -m pytest $ARGS 2>&1 | tail -1)"
done
