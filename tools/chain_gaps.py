"""Where the time of the encode chain goes between its kernels: reads a rocprofv3 kernel trace (…_kernel_trace.csv) of tools/chain_time.py and prints, for the
last runs of the chain with fit (forward -> value sums -> solve -> width sums -> solve -> scan), each kernel's mean duration and the mean idle time in
front of it.   python tools/chain_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("fri::(anonymous namespace)::")[1].split("(")[0][:40]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "fri::" in r["Kernel_Name"]]
# the chain with fit: six kernels starting with fwd_transform and containing two fit_solve kernels
chains, i = [], 0
while i + 6 <= len(ks):
    names = [k[0] for k in ks[i:i + 6]]
    if names[0].startswith("fwd_transform") and sum(n.startswith("fit_solve") for n in names) == 2 and names[5].startswith("predict_histogram"):
        chains.append(ks[i:i + 6])
        i += 6
    else:
        i += 1
chains = chains[-40:]
if not chains:
    sys.exit("no chain with fit in this trace")
print(f"{len(chains)} chains")
for j in range(6):
    dur = sum(c[j][2] - c[j][1] for c in chains) / len(chains) / 1e3
    gap = sum(c[j][1] - (c[j - 1][2] if j else c[j][1]) for c in chains) / len(chains) / 1e3
    print(f"  {chains[0][j][0]:42s} idle before {gap:6.2f} us   runs {dur:7.2f} us")
span = sum(c[5][2] - c[0][1] for c in chains) / len(chains) / 1e3
between = sum(chains[k + 1][0][1] - chains[k][5][2] for k in range(len(chains) - 1)) / max(1, len(chains) - 1) / 1e3
print(f"  first start -> last end {span:7.2f} us; idle between chains {between:6.2f} us")
