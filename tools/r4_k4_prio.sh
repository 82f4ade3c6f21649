#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_TUNING=1
for v in "0 5" "4 5" "4 0" "4 4" "4 3" "0 0" "0 4"; do
  set -- $v
  echo "== FRI_HIP_K4_ABLATE=$1 (4: priority by progress) FRI_HIP_K4_OLDER_EIGHTHS=$2" >> $OUT/prio.log
  FRI_HIP_K4_ABLATE=$1 FRI_HIP_K4_OLDER_EIGHTHS=$2 K2_SLOTS=24 timeout -k 10 200 python3 tools/k2_time.py 2>&1 | grep slots >> $OUT/prio.log
done
cat $OUT/prio.log
