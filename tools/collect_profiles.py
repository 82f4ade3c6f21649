"""Copies the summaries of a tools/profile_round4.sh (round 3: profile_round3.sh) run (gpurun_out/<tag>/) into profiles/ under this round's names and derives
profiles/r03_k1_traffic.json (what bench.py reports as roofline.traffic) from the K1 PMC passes.   python tools/collect_profiles.py <tag> [round]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
copies = {
    "kernel_stats_round3.csv": f"{rnd}_kernel_stats.csv",
    "kernel_stats_round4.csv": f"{rnd}_kernel_stats.csv",
    "kernel_stats_round5.csv": f"{rnd}_kernel_stats.csv",
    "bench_force_dist.json": f"{rnd}_bench_line_force_dist_rccl_one_rank.json",
    "pinned_tiling.txt": f"{rnd}_k1_profiled_tiling.txt",
    "config_report.txt": f"{rnd}_config_report.txt",
    "emit_time.txt": f"{rnd}_emit_time.txt",
    "bench_k20.json": f"{rnd}_bench_line_steps20.json",
    "bench_traced_extras.json": f"{rnd}_bench_line_with_extras_under_rocprofv3.json",
    "pmc_k2_summary.txt": f"{rnd}_k2_pmc_summary.txt",
    "pmc_k2_k4_k3_k5_summary.txt": f"{rnd}_k2_k4_k3_k5_pmc_summary.txt",
    "pmc_k1_summary.txt": f"{rnd}_k1_pmc_summary.txt",
    "bench.json": f"{rnd}_bench_line.json",
    "bench_traced.json": f"{rnd}_bench_line_under_rocprofv3.json",
}
for a, b in copies.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
        print("copied", a, "->", b)
    else:
        print("missing", a)
# K1 plane traffic: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half the bytes of 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM section)
full = open(os.path.join(src, "pmc_k1_summary.txt")).read()
tiling = re.search(r"tiling: ([^:]*):", full)
tiling = tiling.group(1).strip() if tiling else ""
# the batch form: one launch over 24 distinct images, counters per LAUNCH -> per image
if "== K1 batch form" in full:
    bt = full.split("== K1 batch form")[1].split("== K1 RGB")[0]
    bval = lambda name: float(re.search(name + r"\s+n=\s*\d+\s+mean=\s*([0-9.]+)", bt).group(1))
    bf, bw = bval("FETCH_SIZE") / 24, bval("WRITE_SIZE") / 24
    json.dump({
        "kernel": "fwd_transform_quant_kernel<1,false,true,4,true,true,false,false>, grid.y = 24", "workload": "ONE launch over 24 distinct 4096x4096x1 images (fri_hip_transform_quant_batch_dev), per image",
        "tiling": tiling, "bytes_convention": "counter KiB x 1024", "source": f"profiles/{rnd}_k1_pmc_summary.txt", "fetch_size_kb_raw_per_image": round(bf, 1), "write_size_kb_per_image": round(bw, 1),
        "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of 16 B/lane coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE exact for 16 B/lane stores",
        "hbm_bytes_per_launch": int(round((2 * bf + bw) * 1024)), "note": "hbm_bytes_per_launch is per IMAGE here (the key name is the one bench.py reads)",
    }, open(os.path.join(dst, f"{rnd}_k1_batch_traffic.json"), "w"), indent=1)
txt = full.split("== K1 batch form")[0].split("== K1 RGB")[0]
val = lambda name: float(re.search(name + r"\s+n=\s*\d+\s+mean=\s*([0-9.]+)", txt).group(1))
fetch, write = val("FETCH_SIZE"), val("WRITE_SIZE")
out = {
    "kernel": "fwd_transform_quant_kernel<1,false,true,4,true,true,false,false>",
    "workload": "4096x4096x1, single-image launches rotating over 32 slots (every byte from / to HBM)",
    "tiling": tiling,
    "bytes_convention": "counter KiB x 1024",
    "source": f"profiles/{rnd}_k1_pmc_summary.txt",
    "fetch_size_kb_raw": round(fetch, 1),
    "write_size_kb": round(write, 1),
    "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of 16 B/lane coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE exact for 16 B/lane stores",
    "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
}
json.dump(out, open(os.path.join(dst, f"{rnd}_k1_traffic.json"), "w"), indent=1)
print(out)
