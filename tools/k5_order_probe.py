"""What bounds K5's gather (round 5): the gather kernel fed with synthetic stream orders - permutations of the plan's Some nodes that differ only in how many 128-byte lines of the
node-word plane a wave-instruction (64 consecutive symbols) touches and whether the instructions of a wave (1024 consecutive symbols) touch the same lines again.
K5_ORDER = real | incell (see below) | sorted (1 line per instruction) | lines16_reuse (16 lines per instruction, the same 16 in all 16 instructions of a wave) | lines16_fresh (16 lines per
instruction, 256 distinct per wave - what the real order does). Run under rocprofv3 --kernel-trace: the symbol_gather_kernel row is the result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import frave_amd
from frave_amd import emit

kind = os.environ.get("K5_ORDER", "real")
size = int(os.environ.get("K5_SIZE", "4096"))
ctx = frave_amd.Context(0)
P = frave_amd.Plan(ctx, size, size, 1)
real = emit.stream_order(P.centers(), P.valid_mask())
n = len(real)
srt = np.sort(real)
if kind == "real":
    order = real
elif kind == "sorted":
    order = srt
elif kind == "incell":  # the real order with the nodes of every full cell renumbered by their rank in the cell's own part of the stream (what a permuted node-word layout would give)
    cell, heap = real >> 9, real & 511
    idx = np.lexsort((np.arange(n), cell))
    cs = cell[idx]
    start = np.r_[0, np.flatnonzero(np.diff(cs)) + 1]
    cnt = np.diff(np.r_[start, n])
    rank = np.empty(n, np.int64)
    rank[idx] = np.arange(n) - np.repeat(start, cnt)
    full = np.zeros(P.num_cells, bool)
    full[cs[start][cnt == 512]] = True
    order = np.where(full[cell], cell.astype(np.int64) << 9 | rank, real).astype(np.uint32)
else:
    m = n // 1024 * 1024
    if kind == "lines16_reuse":  # window w = sorted block w; instruction k, lane l reads element (l % 16) * 64 + 4 k + l // 16 of the block
        k, l = np.meshgrid(np.arange(16), np.arange(64), indexing="ij")
        idx = ((l % 16) * 64 + 4 * k + l // 16).reshape(-1)
        blocks = srt[:m].reshape(-1, 1024)[:, idx]
    else:  # lines16_fresh: super-blocks of 16 windows = 16384 sorted symbols = 256 lines; window j, instruction k, lane l reads line 16 k + l % 16 of the super-block, halfword 4 j + l // 16
        mm = n // 16384 * 16384
        j, k, l = np.meshgrid(np.arange(16), np.arange(16), np.arange(64), indexing="ij")
        idx = ((16 * k + l % 16) * 64 + 4 * j + l // 16).reshape(-1)
        blocks = np.concatenate([srt[:mm].reshape(-1, 16384)[:, idx].reshape(-1), srt[mm:m]]).reshape(-1, 1024)
    order = np.concatenate([blocks.reshape(-1), srt[m:]])
assert len(np.unique(order)) == n
a = (order[: n // 64 * 64].astype(np.int64) * 2 // 128).reshape(-1, 64)
a.sort(axis=1)
per_instr = (1 + (np.diff(a, axis=1) != 0).sum(axis=1)).mean()
b = (order[: n // 1024 * 1024].astype(np.int64) * 2 // 128).reshape(-1, 1024)
b.sort(axis=1)
per_wave = (1 + (np.diff(b, axis=1) != 0).sum(axis=1)).mean()
P.set_stream_order(order)
F, plane = P.num_cells, P.num_cells * 512
slots = 6
img = np.random.default_rng(1).integers(0, 256, (slots, P.pixel_bytes), dtype=np.uint8)
d_px = torch.from_numpy(img).cuda()
d_co = torch.empty((slots, plane), dtype=torch.int32, device="cuda")
d_w = torch.empty((slots, plane), dtype=torch.uint16, device="cuda")
d_st = torch.empty((slots, n + 64), dtype=torch.uint16, device="cuda")
d_h = torch.empty((slots, 10, 1024), dtype=torch.int32, device="cuda")
d_o = torch.empty((slots,), dtype=torch.int64, device="cuda")
d_par = torch.zeros((slots, 2, 3, 6), dtype=torch.float32, device="cuda")
d_par[:, :, :, 0] = 1.0
s = torch.cuda.current_stream().cuda_stream
for it in range(24):
    k = it % slots
    P.encode_symbols_batch_dev(1, d_px[k].data_ptr(), P.pixel_bytes, None, False, d_par[k].data_ptr(), d_co[k].data_ptr(), plane, d_w[k].data_ptr(), plane, d_st[k].data_ptr(), n + 64,
                               d_h[k].data_ptr(), d_o[k].data_ptr(), stream=s)
torch.cuda.synchronize()
ok = bool((d_w[0].cpu().numpy()[order] == d_st[0, :n].cpu().numpy()).all())
print(f"K5 order {kind}: {per_instr:.1f} lines per wave-instruction, {per_wave:.1f} distinct lines per wave; stream = words[order]: {ok}")
