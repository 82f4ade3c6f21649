export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
for w in "1,1,1,1" "1.1,1.03,0.97,0.9" "1.2,1.07,0.93,0.8" "1.3,1.1,0.9,0.7"; do for s in 16384 8192 2048 1024; do echo -n "size $s weights $w "; SWEEP_SIZE=$s FRI_HIP_RANK_WEIGHTS=$w python tools/k1_sweep.py 32,8,1024 2>&1 | grep band_rows | sed 's/.*}//'; done; done
