#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# Round-5 evidence (round 4's script + the tuned forward tiling, the batch form of K1 as a run and a PMC pass of its own, the instruction-cache counters, the RCCL
# path on one rank), everything in the HBM-bound regime (kernels rotate over enough slots that their inputs cannot come from the 256 MiB Infinity Cache):
# bench.py (plain, and under rocprofv3 --kernel-trace --stats with its extras: one trace holds K1 and every other kernel / chain of the line), K1 RGB and
# 16384^2, K2 / K3 / K4 / K5 standalone over rotating planes, and the PMC counters (separate passes, no trace domains mixed in) of K1 (traffic), K1 RGB,
# K2, K4, K3, K5. All byte counts in the summaries: counter KiB x 1024 (one convention: VERDICT r3).
# usage: tools/profile_round5.sh <tag>   -> gpurun_out/<tag>/...   (then: python3 tools/collect_profiles.py <tag> r05)
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2>> $OUT/bench.err
FRI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_force_dist.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_traced.json 2> $OUT/trace_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench_extras -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_traced_extras.json 2> $OUT/trace_bench_extras.log
echo "bench done" > $OUT/progress.txt
# the forward tiling bench.py's plan measured for itself: the K1 traces and PMC passes below run on the same one (pinned through the tuning knobs)
PIN=$(python3 - <<PY
import json, re
w = json.load(open("$OUT/bench.json"))["config"]["forward_tiling"].get("winner", "")
m = re.match(r"(interleaved|contiguous)/band(\d+)/cells(\d+)", w)
e = []
if m:
    e = [f"FRI_HIP_STRIDED_SHARES={1 if m.group(1) == 'interleaved' else 0}", f"FRI_HIP_BAND_ROWS={m.group(2)}", f"FRI_HIP_CELLS_PER_TILE={m.group(3)}"]
    r = re.search(r"/w([\d.]+)-([\d.]+)", w)
    if r and (r.group(1), r.group(2)) == ("1.40", "0.60"): e.append("FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6")
    if r and (r.group(1), r.group(2)) == ("1.20", "0.80"): e.append("FRI_HIP_RANK_WEIGHTS=1.2,1.05,0.95,0.8")
print(" ".join(e))
PY
)
echo "pinned forward tiling: $PIN" | tee $OUT/pinned_tiling.txt
K1_SIZE=4096 K1_SLOTS=24 K1_BATCH=24 env $PIN rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_batch -- python3 $R/tools/k1_run.py 12 > $OUT/trace_k1_batch.log 2>&1
K1_SIZE=4096 SWEEP_C=3 K1_SLOTS=12 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_c3 -- python3 $R/tools/k1_run.py 240 > $OUT/trace_k1_c3.log 2>&1
K1_SIZE=16384 SWEEP_C=1 K1_SLOTS=2 K1_SPIN_UP=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_16k -- python3 $R/tools/k1_run.py 40 > $OUT/trace_k1_16k.log 2>&1
K2_TRUSTED=1 K2_SLOTS=12 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4.log 2>&1
K2_TRUSTED=1 K2_SLOTS=4 SWEEP_C=3 K5=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4_c3 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4_c3.log 2>&1
K2_TRUSTED=1 K2_SIZE=16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4_16k -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4_16k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chain -- python3 $R/tools/chain_hbm.py > $OUT/trace_chain.log 2>&1
echo "traces done" >> $OUT/progress.txt
pass() { dir=$1; shift; script=$1; shift; K2_TRUSTED=1 K2_SLOTS=12 K1_SPIN_UP=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$dir -- python3 $R/tools/$script > $OUT/$dir.log 2>&1; echo "$dir" >> $OUT/progress.txt; }
# K2 / K3 / K4 / K5 (tools/k2_time.py launches each 21 times, rotating over 12 coefficient planes)
pass sq1 k2_time.py SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass sq2 k2_time.py SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcc1 k2_time.py FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 k2_time.py WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# instruction cache (K2's first tile)
pass ic1 k2_time.py SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
# K1 plane on the pinned (= the bench's measured) tiling: traffic for bench.py's roofline.traffic, 32 rotating slots; the batch form (24 distinct images per launch); K1 RGB, 12 slots
export K1_SLOTS=32 $PIN
pass k1_fetch "k1_run.py 64" FETCH_SIZE GRBM_GUI_ACTIVE
pass k1_write "k1_run.py 64" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass k1_sq1 "k1_run.py 64" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass k1_sq2 "k1_run.py 64" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
export K1_SLOTS=24 K1_BATCH=24
pass k1b_fetch "k1_run.py 6" FETCH_SIZE GRBM_GUI_ACTIVE
pass k1b_write "k1_run.py 6" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
unset K1_BATCH
for v in $PIN; do unset ${v%%=*}; done
export SWEEP_C=3 K1_SLOTS=12
pass rgb_sq2 "k1_run.py 48" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass rgb_fetch "k1_run.py 48" FETCH_SIZE GRBM_GUI_ACTIVE
pass rgb_write "k1_run.py 48" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
unset SWEEP_C K1_SLOTS
cd $R
python3 tools/kernel_stats_from_traces.py $OUT > $OUT/kernel_stats_round5.csv
cat $OUT/kernel_stats_round5.csv
for needle in "predict_histogram_kernel3<true, false, false, false>" "predict_histogram_kernel3<false, false, false, false>" "predict_histogram_kernel3<false, true, false, false>" "fit_accumulate_kernel2<0" "fit_accumulate_kernel2<1" inverse_transform symbol_gather symbol_stream; do
  echo "== $needle: mean per launch (FETCH_SIZE / WRITE_SIZE in KiB) =="
  for p in sq1 sq2 tcc1 tcc2 ic1; do python3 tools/pmc_summary.py $OUT/$p "$needle"; done
done > $OUT/pmc_k2_k4_k3_k5_summary.txt
{ echo "== K1 plane (fwd_transform_quant_kernel<1,...>), 4096x4096x1 over 32 rotating slots, tiling: $PIN: mean per launch (FETCH_SIZE / WRITE_SIZE in KiB) =="; for p in k1_fetch k1_write k1_sq1 k1_sq2; do python3 tools/pmc_summary.py $OUT/$p fwd_transform; done;
  echo "== K1 batch form: ONE launch over 24 distinct 4096x4096x1 images (grid.y = 24), same tiling: mean per LAUNCH (divide by 24 for an image) =="; for p in k1b_fetch k1b_write; do python3 tools/pmc_summary.py $OUT/$p fwd_transform; done;
  echo "== K1 RGB (fwd_transform_quant_kernel<3,...>), 4096x4096x3 over 12 rotating slots =="; for p in rgb_sq2 rgb_fetch rgb_write; do python3 tools/pmc_summary.py $OUT/$p fwd_transform; done; } > $OUT/pmc_k1_summary.txt
cat $OUT/pmc_k2_k4_k3_k5_summary.txt $OUT/pmc_k1_summary.txt
cat $OUT/bench.json $OUT/bench_k20.json $OUT/bench_force_dist.json
grep -h "us/launch\|slots=\|CHAIN" $OUT/trace_*.log
# the host emitter on this box's CPU (array route against stream route) and the five-config report
python3 tests/tools/emit_time.py 4096 4096 1 > $OUT/emit_time.txt 2>&1
./frave_amd/host/fri_driver batch-frv 4096 4096 1 24 --emitters 8 >> $OUT/emit_time.txt 2>&1
python3 tests/tools/report_configs.py > $OUT/config_report.txt 2>&1
tail -5 $OUT/emit_time.txt
