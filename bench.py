#!/usr/bin/env python3
"""bench.py -- Mpixels/s of libfri's encode hot path (transform + quantisation) at 4096x4096 on MI355X.

A "step" is one pass of K1 (address map + residue transform + quantiser) over one synthetic 8-bit
4096x4096 plane per GPU, input already resident in HBM. Steps rotate over enough distinct image/coefficient
slots to exceed the 256 MiB Infinity Cache, so the timed traffic really goes to HBM. Ranks hold independent
images (one process per GPU, no data-path collective): weak scaling.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W = H = 4096
CHANNELS = 1
SPIN_UP_LAUNCHES = 4000  # ~70 ms of untimed work before the warm-up steps, see main()
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def measured_traffic():
    """HBM bytes per K1 launch from the committed rocprofv3 PMC pass (profiles/), or None. bench.py cannot collect
    hardware counters itself; the number is tied to the kernel named in the file."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_k1_traffic.json")) as f:
            return int(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or usable, usable


def cpu_baseline(seconds_budget=12.0):
    """The CPU oracle (a port, not libfri itself: no Rust toolchain here) on a bounded sample of the same workload.

    libfri is single-threaded per image; a batch is embarrassingly parallel over images, so the all-core
    figure runs one oracle instance per usable host core on independent images (ctypes releases the GIL during the C call).
    `value`/`cores` are the all-core figure, `value_1thread` the single-thread one."""
    import numpy as np

    from oracle import fri_oracle
    from tests.common import gen_image

    ones = np.ones(32, np.int32)
    model, host_cores, usable = _cpu_info()

    def one(img):
        wl = fri_oracle.Wavelet(img, H, W, CHANNELS)  # from_raster: lattice + address map + residue transform
        wl.quantize(ones)
        wl.close()

    imgs = [gen_image("noise", W, H, CHANNELS, 1000 + k) for k in range(3)]
    fri_oracle.lib()
    n1, t1 = 0, 0.0
    while n1 < 3 and t1 < seconds_budget / 2:
        t0 = time.perf_counter()
        one(imgs[n1])
        t1 += time.perf_counter() - t0
        n1 += 1
    per_image = t1 / n1
    # One forked worker process per usable core (threads of one process serialise on the address-space lock while the oracle's
    # hash maps fault their pages in: 2.3x on 8 cores against 5.2x with processes). Forked, not spawned, and before this process
    # has touched the GPU (main() calls this first). An oracle instance of a 4096^2 plane holds ~2.5 GB, hence the cap.
    import multiprocessing as mp

    workers = max(1, min(usable, 16))  # a one-GPU box's CPU share is 16 cores, whatever the host shows
    rounds = max(1, min(3, int(seconds_budget / 2 / max(per_image * 1.5, 1e-3))))

    def work(k):
        for r in range(rounds):
            one(imgs[(k + r) % len(imgs)])

    fork = mp.get_context("fork")
    procs = [fork.Process(target=work, args=(k,)) for k in range(workers)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    tn = time.perf_counter() - t0
    if any(p.exitcode != 0 for p in procs):
        raise RuntimeError("cpu_baseline worker failed")
    return {
        "value": round(workers * rounds * W * H / tn / 1e6, 3),
        "unit": "Mpixels/s",
        "cores": workers,
        "kind": "port",
        "value_1thread": round(n1 * W * H / t1 / 1e6, 3),
        "host_cores": host_cores,
        "host_cores_usable": usable,
        "cpu_model": model,
        "sample": f"oracle/fri_oracle.c (C restatement of libfri, transform+quant of {W}x{H}x{CHANNELS} noise planes): {workers} processes x {rounds} image(s) "
                  f"in {tn:.1f} s (one single-threaded instance per core, like running libfri per image); 1 thread: {n1} image(s) in {t1:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)  # on top of the untimed spin-up, see SPIN_UP_LAUNCHES
    ap.add_argument("--slots", type=int, default=8, help="distinct image/coefficient buffer pairs the steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time an 8-image batch launch of K1, K2 (predict+histogram) and K3 (inverse)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the CPU leg runs first: it forks workers, which must happen before this process initialises the GPU
    cpu = cpu_baseline() if rank == 0 and world == 1 and not args.no_cpu_baseline else None

    import numpy as np
    import torch

    import frave_amd

    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N>1", file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("FRI_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path with a single rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = frave_amd.Context(local_rank)  # raises if the HIP library or a gfx950 GPU is missing: no fallback
    plan = frave_amd.Plan(ctx, W, H, CHANNELS)
    F = plan.num_cells
    alg_bytes = plan.pixel_bytes + plan.coef_count * 4  # u8 read once + int32 coefficient write

    # The job's batch is world x (warmup + steps) images; image i belongs to rank i mod world (the library's partition,
    # fri_hip_shard_image - what fri_hip_multi_transform_quant and `fri_driver batch --gpus N` use), no data-path collective.
    # A rank's k-th step transforms its k-th image; image content depends on the global index only. The rank keeps
    # `slots` of its images resident and the steps rotate over them.
    from frave_amd.dist import images_for_rank

    mine = images_for_rank(world * (args.warmup + args.steps), rank, world)
    assert len(mine) == args.warmup + args.steps and all(i % world == rank for i in mine)
    d_px = torch.empty((args.slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
    for k in range(args.slots):
        gen = torch.Generator(device="cuda").manual_seed(0xF7A5E000 + mine[k % len(mine)])
        d_px[k].copy_(torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda", generator=gen))
    d_co = torch.empty((args.slots, plan.coef_count), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    px0, co0 = d_px.data_ptr(), d_co.data_ptr()
    pstride, cstride = plan.pixel_bytes, plan.coef_count

    def step(i):
        k = i % args.slots
        plan.transform_quant_dev(px0 + k * pstride, co0 + k * cstride * 4, stream=stream)

    from frave_amd.dist import timed_region

    # Untimed spin-up before the W warm-up steps: a fresh GPU takes tens of milliseconds of work to reach its steady clocks and
    # warm translations (measured on MI355X: 17.5-18.3 us per launch after 40 steps, 17.2-17.3 us after 4000), and the metric is the
    # steady-state rate of a device that is kept busy. SPIN_UP launches of the same kernel on the same slots, never timed.
    plan.time_transform_quant_dev(args.slots, px0, pstride, co0, cstride, SPIN_UP_LAUNCHES, stream=stream)
    for i in range(args.warmup):
        step(i)
    if dist is not None:  # the first collectives of a process build the communicator (milliseconds, and its threads stay busy a little longer): not inside the fence of the timed region
        for _ in range(3):
            dist.barrier()
        torch.cuda.synchronize()

    timed = {}

    def timed_steps():
        # K steps = K single-image launches over the rotating slots, issued by the library's native loop (what a C++ / Rust host
        # does through the same ABI; a Python call per step adds ~1 us of host time between launches). The same call brackets
        # the K launches with HIP events on the launch stream: the dominant kernel's mean launch period over the timed region.
        t_call = time.perf_counter()
        timed["kernel_us"] = plan.time_transform_quant_dev(args.slots, px0, pstride, co0, cstride, args.steps, stream=stream)
        timed["call_us"] = (time.perf_counter() - t_call) * 1e6

    # barrier + device synchronisation on both sides, MAX over ranks (the protocol tests/test_multi_gloo.py exercises on gloo)
    elapsed = timed_region(timed_steps, dist=dist, device_sync=torch.cuda.synchronize, device="cuda")
    kernel_us = timed["kernel_us"]
    achieved = alg_bytes / (kernel_us * 1e-6) / 1e9

    out = {
        "metric": "Mpixels/s encode (transform+quant) at 4096x4096",
        "value": round(world * args.steps * W * H / elapsed / 1e6, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8->i32",
        "data": "synthetic",
        "config": {
            "workload": f"1 x {W}x{H} 8-bit plane per GPU per step (BASELINE config 2), K1 = address map + residue transform + quant, "
                        f"F={F} cells, {args.slots} rotating HBM-resident slots",
            "channels": CHANNELS,
            "spin_up_launches": SPIN_UP_LAUNCHES,
            "parallelism": f"batch sharded by image, image i -> GPU i mod {world} (fri_hip_shard_image), no collective",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(),
            "kernel": "fwd_transform_quant_kernel<1,false,true,4,true>",
            "kernel_us": round(kernel_us, 3),
            "algorithmic_bytes_per_launch": alg_bytes,
        },
        # where the wall clock of the timed region goes (rank 0): the K launches by HIP events, what the native call adds before the first
        # and after the last of them, and the closing synchronise - a fixed 40-90 us that K = 20 feels and K = 400 does not
        "timed_region": {
            "wall_us": round(elapsed * 1e6, 1),
            "kernels_us": round(kernel_us * args.steps, 1),
            "call_us": round(timed["call_us"], 1),
            "after_call_us": round(elapsed * 1e6 - timed["call_us"], 1),
        },
    }

    # The same kernel with all slots in ONE launch (the batch entry point; BASELINE config 4 runs like this): the ramp and the
    # tail of consecutive images overlap. Reported next to the single-image figures, never as `value`; only with --extras, so that
    # the default command launches nothing but single-image kernels (its rocprofv3 kernel stats are then those of `roofline`).
    if args.extras and rank == 0:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        plan.transform_quant_dev(px0, co0, stream=stream, n_images=args.slots, pixel_stride=pstride, coef_stride=cstride)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            plan.transform_quant_dev(px0, co0, stream=stream, n_images=args.slots, pixel_stride=pstride, coef_stride=cstride)
        ev1.record()
        torch.cuda.synchronize()
        us_img = ev0.elapsed_time(ev1) / reps / args.slots * 1e3
        out["batch_launch"] = {
            "images_per_launch": args.slots,
            "us_per_image": round(us_img, 3),
            "achieved_GBps": round(alg_bytes / us_img / 1e3, 1),
            "frac_of_peak": round(alg_bytes / us_img / 1e3 / HBM_PEAK_GBS, 4),
        }

    if args.extras and rank == 0:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
        wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
        d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
        d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
        d_o = torch.empty(1, dtype=torch.int64, device="cuda")
        d_back = torch.empty(plan.pixel_bytes, dtype=torch.uint8, device="cuda")
        reps = 20

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / reps * 1e3

        k2_us = timed(lambda: plan.predict_histogram_dev(co0, 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=stream))
        k3_us = timed(lambda: plan.inverse_transform_dev(co0, d_back.data_ptr(), stream=stream))
        assert torch.equal(d_back, d_px[0]), "K3(K1(x)) != x"
        d_g = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
        d_w = torch.empty(18, dtype=torch.float64, device="cuda")
        k4v_us = timed(lambda: plan.fit_value_sums_dev(co0, 0, d_g.data_ptr(), stream=stream))
        k4w_us = timed(lambda: plan.fit_width_sums_dev(co0, 0, vp, d_g.data_ptr(), d_w.data_ptr(), stream=stream))
        # the whole device part of FRIEncoder::encode for one image, coefficients staying in HBM (fri_hip_encode_image_dev): with the
        # parameters given (K1 -> K2) and with the fit (K1 -> fit sums -> solve -> fit sums -> solve -> K2; two host round trips inside)
        vp3, wp3 = vp.reshape(1, 3, 6).copy(), wp.reshape(1, 3, 6).copy()
        enc_given = timed(lambda: plan.encode_image_dev(px0, co0, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), vp3, wp3, fit=False, stream=stream))
        vpf, wpf = np.zeros((1, 3, 6), np.float32), np.zeros((1, 3, 6), np.float32)
        enc_fit = timed(lambda: plan.encode_image_dev(px0, co0, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), vpf, wpf, fit=True, stream=stream))
        k2_bytes = F * 512 * (4 + 1 + 4) + 10 * 1024 * 4  # coefficient read + bucket + prediction write + histogram
        out["extras"] = {
            "k2_predict_histogram_us": round(k2_us, 2),
            "k2_Mpixels_per_s": round(W * H / k2_us, 1),
            "k2_algorithmic_GBps": round(k2_bytes / k2_us / 1e3, 1),
            "k2_note": "through fri_hip_predict_histogram_dev, i.e. the fast kernel + the exact int32 kernel that returns at once for the forward kernel's output",
            "k3_inverse_us": round(k3_us, 2),
            "k3_Mpixels_per_s": round(W * H / k3_us, 1),
            "k3_algorithmic_GBps": round(alg_bytes / k3_us / 1e3, 1),
            "k4_fit_value_sums_us": round(k4v_us, 2),
            "k4_fit_width_sums_us": round(k4w_us, 2),
            "encode_device_us": {"parameters_given": round(enc_given, 2), "with_fit": round(enc_fit, 2)},
            "encode_pcie_bytes_per_image": {"host_to_device": plan.pixel_bytes, "device_to_host": F * 512 * (4 + 1 + 4) + 10 * 1024 * 4 + 8,
                                            "note": "fri_hip_encode_image: the pixels go up once, coefficients / bucket / prediction / histogram come down once; nothing is uploaded twice"},
            "note": "per channel plane; K2 = 6-neighbour gather + bucket/prediction + histogram, K3 = dequant + inverse transform, K4 = normal-equation sums of the predictor fit",
        }

    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
