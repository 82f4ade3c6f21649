export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# K1 at 4096^2 for dispatch-rank share weights; interleaved repeats on one box.
for rep in 1 2 3; do
  for w in "1,1,1,1" "1.3,1.1,0.9,0.7" "1.24,1.0,1.0,0.76" "1.24,1.12,0.88,0.76" "1.2,1.1,0.9,0.8" "1.35,1.05,0.9,0.7" "1.3,1.15,0.85,0.7"; do
    echo -n "weights $w  "; FRI_HIP_RANK_WEIGHTS=$w timeout -k 10 120 python tools/k1_run.py 300 2>&1 | grep K1
  done
done
