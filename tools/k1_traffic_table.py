"""Merge a traffic_by_tiling.txt of tools/r5_k1_traffic.sh into profiles/r05_k1_traffic_by_tiling.json (+ .txt): per tiling FETCH_SIZE x 2 (the gfx950 correction for
16 B/lane loads, MI355X_MICROARCH.md) + WRITE_SIZE, counter KiB x 1024. usage: k1_traffic_table.py gpurun_out/<tag>/traffic_by_tiling.txt"""
import json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
jp = os.path.join(root, "profiles", "r05_k1_traffic_by_tiling.json")
tp = os.path.join(root, "profiles", "r05_k1_traffic_by_tiling.txt")
table = json.load(open(jp))
alg = table["algorithmic_bytes"]
text = open(sys.argv[1]).read()
have = open(tp).read()
for block in re.split(r"^== ", text, flags=re.M)[1:]:
    name = block.split("\n", 1)[0].strip()
    f = re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=\s*([\d.]+)", block)
    w = re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=\s*([\d.]+)", block)
    if not (f and w):
        print("incomplete:", name)
        continue
    m = re.match(r"(interleaved|contiguous)_band(\d+)_cells(\d+)", name)
    key = f"{m.group(1)}/band{m.group(2)}/cells{m.group(3)}"
    fetch, write = float(f.group(1)), float(w.group(1))
    hbm = int(round((2 * fetch + write) * 1024))
    table["by_tiling"][key] = {"fetch_size_kb_raw": fetch, "write_size_kb": write, "hbm_bytes_per_launch": hbm, "of_algorithmic": round(hbm / alg, 4)}
    if f"== {name}\n" not in have:
        have += f"== {name}\n" + "\n".join(l for l in block.split("\n")[1:] if l.strip()) + "\n"
    print(key, hbm, round(hbm / alg, 4))
json.dump(table, open(jp, "w"), indent=1)
open(tp, "w").write(have)
