#!/usr/bin/env python3
"""Search for the intra-cell LDS layout of the gather kernels (K2 predict + histogram, K4 fit sums).

A tile's 36 cells sit in LDS as 16-bit values, 1024 B per cell. The 6-neighbour gather of
ContextModeler::get_neighbour_values (context_modeling.rs:25-77) reads, for the 32 lanes of one LDS lane group, 32 halfwords
whose positions are fixed by the image-independent neighbour table - so which LDS banks collide is a property of WHERE inside
its cell's 1 KiB each node is stored. With the nodes in heap order the 48 gather instructions of a cell take 141-178 LDS cycles
instead of the 96 of a conflict-free gather (ds_read_u16: two groups of 32 lanes, bank = (address / 4) mod 32, one cycle per
distinct dword on the busiest bank; /opt/skills/guides/MI355X_MICROARCH.md, LDS).

This tool anneals a permutation of the 256 halfword PAIRS of a cell (pairs stay together so that staging writes whole dwords;
pairs stay inside their tree level's region) for the kernels' lane map (group_nodes below) and writes
frave_amd/csrc/gather_layout.inc. Cost = LDS cycles of the 48 gathers of a cell + what the staging writes (ds_write_b32,
patterns in Layout.__init__) lose beyond their 2-way free conflicts.

    python tools/lds_layout_search.py [--iters N] [--seed S] [--write]
"""
import argparse
import math
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIDE = 6          # slots per tile edge (4 x 4 block + halo ring)
SLOT_HW = 512     # halfwords per slot
DA = [0, 1, 1, 0, -1, -1, 0]   # lattice deltas of the neighbour-cell list {self, +V9[0..5]} (gather_common.hpp)
DB = [0, 0, -1, -1, 0, 1, 1]


def load_table():
    import frave_amd as fa

    tab = fa.Plan(None, 512, 512, 1).neighbour_table()
    return [[(int(e) & 511, (int(e) >> 9) & 7, (int(e) >> 15) & 1) for e in row] for row in tab]


def group_nodes(g):
    """LDS lane group g = (role, n, half): lanes 32 half .. 32 half + 31 of gather instruction n of a wave of that role.
    Level-8 waves (role 1): lane L handles heap nodes 256 + 4 L + n. Waves of levels 0..7 (role 0): nodes 2 L, 2 L + 1 (levels 0..6)
    and 128 + 2 L, 128 + 2 L + 1 (level 7), so that every instruction works on one parameter group."""
    role, n, half = g >> 3, (g >> 1) & 3, g & 1
    lanes = range(32 * half, 32 * half + 32)
    if role:
        return [256 + 4 * lane + n for lane in lanes]
    return [128 * (n >> 1) + 2 * lane + (n & 1) for lane in lanes]


class Layout:
    def __init__(self, table, slot_dwords=SLOT_HW // 2, fit_writes=False):
        """slot_dwords: distance of two cells in LDS (K2: 256 = 1 KiB; K4: 260 - its cells carry their zero words behind them, so neighbouring cells are
        four banks apart). fit_writes: K4's staging writes (lane L writes pairs 4 L + w, w = 0..3) instead of K2's."""
        self.slot_dwords = slot_dwords
        self.pair_pos = list(range(256))  # pair q = heap nodes 2q, 2q+1 -> dword position inside the slot
        self.desc = []                    # [group][k] -> list of (slot offset in dwords, pair, or -1 = "never a node": the zero word)
        for g in range(16):
            row = []
            for k in range(6):
                lst = []
                for p in group_nodes(g):
                    h, s, nv = table[p][k]
                    lst.append((None, -1) if nv else ((DA[s] * SIDE + DB[s]) * slot_dwords, h >> 1))
                row.append(lst)
            self.desc.append(row)
        # staging writes (ds_write_b32, one pair per lane and instruction, two lane groups each). Halo cells: instruction w of a wave
        # writes pair {2L, 2L+1, 128+2L, 128+2L+1}[w] of lane L (w = 0..3; role-1 waves stage their own cells with w = 2, 3);
        # role-0 waves stage pairs L and 64 + L of their own cells (w = 4, 5); quarter cells: pair 64 q + L (w = 6..9, the first two
        # coincide with w = 4, 5)
        def wpairs(w, lane):
            if fit_writes:
                return 4 * lane + (w & 3)
            return 2 * lane + (w & 1) + 128 * (w >> 1) if w < 4 else 64 * (w - 4) + lane
        self.wdesc = [[[wpairs(w, lane) for lane in range(32 * half, 32 * half + 32)] for half in range(2)] for w in range(8)]
        self.refs = [set() for _ in range(256)]
        for g in range(16):
            for k in range(6):
                for so, q in self.desc[g][k]:
                    if q >= 0:
                        self.refs[q].add(("g", g, k))
        for w in range(8):
            for half in range(2):
                for q in self.wdesc[w][half]:
                    self.refs[q].add(("w", w, half))

    def cost(self, key):
        banks = {}
        if key[0] == "g":
            for so, q in self.desc[key[1]][key[2]]:
                a = 10 ** 6 if q < 0 else so + self.pair_pos[q] + 100 * self.slot_dwords * 32
                banks.setdefault(a & 31, set()).add(a)
            mx = max(len(s) for s in banks.values())
            return mx + 0.02 * sum(len(s) - 1 for s in banks.values())
        for q in self.wdesc[key[1]][key[2]]:
            a = self.pair_pos[q]
            banks.setdefault(a & 31, set()).add(a)
        mx = max(len(s) for s in banks.values())
        return 2.25 * max(0, mx - 2) + 0.01 * sum(len(s) - 1 for s in banks.values())  # 2.25 cells staged per cell predicted

    def all_keys(self):
        return [("g", g, k) for g in range(16) for k in range(6)] + [("w", w, h) for w in range(8) for h in range(2)]


def level_of_pair(q):
    return max(2 * q, 1).bit_length() - 1


def anneal(lay, iters, seed, log=True):
    rnd = random.Random(seed)
    c = {key: lay.cost(key) for key in lay.all_keys()}
    tot = sum(c.values())
    t0, t1 = 0.5, 0.01
    for it in range(iters):
        temp = t0 * (t1 / t0) ** (it / iters)
        u = rnd.randrange(1, 256)
        lv = level_of_pair(u)
        lo, hi = max((1 << lv) // 2, 1), (2 << lv) // 2
        if hi - lo < 2:
            continue
        v = rnd.randrange(lo, hi)
        if v == u:
            continue
        aff = lay.refs[u] | lay.refs[v]
        old = sum(c[k] for k in aff)
        lay.pair_pos[u], lay.pair_pos[v] = lay.pair_pos[v], lay.pair_pos[u]
        new = {k: lay.cost(k) for k in aff}
        d = sum(new.values()) - old
        if d <= 0 or rnd.random() < math.exp(-d / temp):
            c.update(new)
            tot += d
        else:
            lay.pair_pos[u], lay.pair_pos[v] = lay.pair_pos[v], lay.pair_pos[u]
        if log and it % 100000 == 0:
            print(f"  iter {it}: cost {tot:.2f}", flush=True)
    cycles = sum(int(c[("g", g, k)]) for g in range(16) for k in range(6))
    wmax = max(int(round(c[("w", w, h)] / 2.25)) + 2 for w in range(8) for h in range(2))
    return cycles, wmax


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--k4", action="store_true", help="K4's geometry: cells 1040 B apart, its staging writes; evaluates heap order, K2's layout and an annealed one (never written)")
    args = ap.parse_args()
    table = load_table()
    if args.k4:
        lay = Layout(table, 260, True)
        gathers = lambda l: sum(int(l.cost(("g", g, k))) for g in range(16) for k in range(6))
        print(f"K4 geometry, heap order: {gathers(lay)} LDS cycles for the 48 gather instructions of a cell (conflict-free: 96)")
        inc = open(os.path.join(ROOT, "frave_amd", "csrc", "gather_layout.inc")).read()
        lay.pair_pos = [int(v) for v in "".join(l for l in inc.splitlines() if not l.startswith("//")).replace(" ", "").split(",") if v]
        print(f"K4 geometry, K2's layout: {gathers(lay)} LDS cycles")
        lay.pair_pos = list(range(256))
        cycles, wmax = anneal(lay, args.iters, args.seed)
        print(f"K4 geometry, annealed for it: {cycles} LDS cycles; staging writes at most {wmax}-way on a bank")
        return
    lay = Layout(table)
    base = sum(int(lay.cost(("g", g, k))) for g in range(16) for k in range(6))
    print(f"heap order: {base} LDS cycles for the 48 gather instructions of a cell (conflict-free: 96)")
    cycles, wmax = anneal(lay, args.iters, args.seed)
    assert sorted(lay.pair_pos) == list(range(256))
    assert all(level_of_pair(q) == level_of_pair(lay.pair_pos[q]) for q in range(1, 256)) and lay.pair_pos[0] == 0
    print(f"annealed:   {cycles} LDS cycles; staging writes at most {wmax}-way on a bank")
    if args.write:
        path = os.path.join(ROOT, "frave_amd", "csrc", "gather_layout.inc")
        with open(path, "w") as f:
            f.write("// gather_layout.inc -- generated by tools/lds_layout_search.py (--iters %d --seed %d): dword position, inside a cell's 1 KiB LDS slot, of the\n"
                    "// halfword pair (heap nodes 2q, 2q + 1). A permutation of 0..255 that keeps every pair inside its tree level's region.\n"
                    "// %d LDS cycles for the 48 gather instructions of a cell (heap order: %d, conflict-free: 96).\n" % (args.iters, args.seed, cycles, base))
            for r in range(0, 256, 16):
                f.write(", ".join(f"{v:3d}" for v in lay.pair_pos[r:r + 16]) + ",\n")
        print("wrote", path)


if __name__ == "__main__":
    main()
