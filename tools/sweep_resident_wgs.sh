export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
for rep in 1 2 3; do
 echo -n "1024 default           "; python tools/k1_run.py 300 | grep K1
 for w in "1,1,1,1" "1.2,1.0,0.8,0.8" "1.3,1.0,0.7,0.7" "1.25,1.05,0.7,0.7"; do echo -n "768 weights $w  "; FRI_HIP_TARGET_WGS=768 FRI_HIP_RANK_WEIGHTS=$w python tools/k1_run.py 300 | grep K1; done
 for w in "1.3,0.7,1,1" "1.2,0.8,1,1"; do echo -n "512 weights $w  "; FRI_HIP_TARGET_WGS=512 FRI_HIP_RANK_WEIGHTS=$w python tools/k1_run.py 300 | grep K1; done
done
