for rep in 1 2; do for f in 0 1 2 3 4 5; do echo -n "store=$f "; FRI_HIP_LIBRARY=$PWD/build_variants/libfri_st$f.so timeout -k 10 120 python tools/k1_run.py 300 2>&1 | grep K1; done; done
for f in 1 3 4; do echo "batch store=$f"; FRI_HIP_LIBRARY=$PWD/build_variants/libfri_st$f.so timeout -k 10 120 python tools/k1_batch.py 2>&1 | tail -3; done
