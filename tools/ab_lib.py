"""Interleaved A/B of builds of libfri_hip.so on ONE box (the boxes of the pool differ by more than most effects).

    python3 tools/ab_lib.py [rounds] A.so B.so [C.so ...]       ("-" = the in-tree build)

Each round runs tools/ab_probe.py once per library (a fresh process each: FRI_HIP_LIBRARY selects the build) and the medians over the
rounds are printed: K1, K2 behind the generic entry point (fast kernel + exact-kernel guard), the chain K1 -> K2 with given parameters
(K2 alone = chain - K1), K4 value / width sums, K3, the chain with the fit. AB_SIZE / AB_C select the image (default 4096, 1)."""
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
        if lib != "-":
            env["FRI_HIP_LIBRARY"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(HERE, "ab_probe.py")], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(f"{lib}: probe failed\n{out.stdout[-400:]}\n{out.stderr[-800:]}")
            continue
        res[lib].append([float(x) for x in line[0].split()[1:]])
names = ["k1", "k2+guard", "k1->k2", "k2", "k4v", "k4w", "k3", "chain_fit"]
print("library".ljust(44) + "".join(n.rjust(10) for n in names) + "   [us, median of %d]" % rounds)
for lib in libs:
    if res[lib]:
        med = [statistics.median(c) for c in zip(*res[lib])]
        print(lib[-43:].ljust(44) + "".join(f"{v:10.2f}" for v in med))
