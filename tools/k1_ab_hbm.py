"""Interleaved A/B of builds of libfri_hip.so for K1 in the HBM-bound regime, on ONE box: each round runs tools/k1_probe_hbm.py once per library
(a fresh process each; "-" = the in-tree build; a library may carry environment settings: "lib.so:AB_TUNE=1,FRI_HIP_BAND_ROWS=72").
    python3 tools/k1_ab_hbm.py [rounds] A.so B.so ...        (AB_W / AB_H / AB_C / AB_STREAMS select the image and the extra column)"""
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
libs = args or ["-"]
res = {lib: [] for lib in libs}
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        path, _, extra = lib.partition(":")
        if path != "-":
            env["FRI_HIP_LIBRARY"] = os.path.abspath(path)
        import re

        for kv in filter(None, re.split(r",(?=[A-Z][A-Z0-9_]*=)", extra)):  # (a value may hold commas: FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6)
            k, v = kv.split("=", 1)
            env[k] = v
            if k.startswith("FRI_HIP_"):
                env["FRI_HIP_TUNING"] = "1"
        out = subprocess.run([sys.executable, os.path.join(HERE, "k1_probe_hbm.py")], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(f"{lib}: probe failed\n{out.stdout[-400:]}\n{out.stderr[-800:]}")
            continue
        if r == 0 and "tune:" in out.stderr:
            print(lib, [l for l in out.stderr.splitlines() if l.startswith("tune:")][0])
        res[lib].append([float(x) for x in line[0].split()[1:]])
print(f"W={os.environ.get('AB_W', '4096')} H={os.environ.get('AB_H', '4096')} C={os.environ.get('AB_C', '1')}: us per launch (median of {rounds} processes), then the n-stream period if asked")
for lib in libs:
    if res[lib]:
        med = [statistics.median(c) for c in zip(*res[lib])]
        print(lib[-60:].ljust(62) + "".join(f"{v:10.3f}" for v in med) + "   all: " + " ".join(f"{x[0]:.2f}" for x in res[lib]))
