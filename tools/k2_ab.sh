#!/bin/bash
# usage: tools/k2_ab.sh <rounds> <lib or -> ...   interleaved, one process per library and round
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then echo -n "in-tree            "; python3 tools/k2_ab.py 2>&1 | grep K2AB; else echo -n "$(basename $lib) "; FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib python3 tools/k2_ab.py 2>&1 | grep K2AB; fi
  done
done
