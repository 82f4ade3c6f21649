#!/usr/bin/env python3
"""bench.py -- Mpixels/s of libfri's encode hot path (transform + quantisation) at 4096x4096 on MI355X.

A "step" is one pass of K1 (address map + residue transform + quantiser) over one synthetic 8-bit
4096x4096 plane per GPU, input already resident in HBM. Steps rotate over 24 distinct image/coefficient slots
(403 MB of pixels, 2 GB with the coefficients): with 8 slots (rounds 1-3) the 134 MB of pixels stayed in the 256 MiB
Infinity Cache and the launch period was 16.5 us; from 24 slots on it is flat at the HBM-bound figure (tools/k1_slots.py,
DESIGN.md section 7). Ranks hold independent images (one process per GPU, no data-path collective): weak scaling.

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


def measured_traffic(name="k1_traffic", tiling=None):
    """HBM bytes per K1 launch (per image for the batch form) from the committed rocprofv3 PMC passes (profiles/), or None. bench.py cannot collect
    hardware counters itself; the number is tied to the kernel and tiling named in the file (newest round first). `tiling`: the label of the forward
    tiling the plan measured for itself ("contiguous/band72/cells8/..."): the single-launch figure is then the one of THAT tiling, if it was profiled
    (profiles/r05_k1_traffic_by_tiling.json: 1.017x on the contiguous 72-row tiling, 1.047-1.054x on the interleaved 16-row ones)."""
    if tiling:  # a tiling that was not profiled has no measured traffic: null, never another tiling's number
        try:
            with open(os.path.join(ROOT, "profiles", "r05_k1_traffic_by_tiling.json")) as f:
                by = json.load(f)["by_tiling" if name == "k1_traffic" else "batch_form_by_tiling"]
            key = "/".join(tiling.split("/")[:3])
            return int(by[key]["hbm_bytes_per_launch"]) if key in by else None
        except Exception:
            return None
    for rnd in ("r05", "r04"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")) as f:
                return int(json.load(f)["hbm_bytes_per_launch"])
        except Exception:
            continue
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

    # A one-GPU box of the pool is a 1/8 share of an 8-GPU host: its CPU share is 16 cores whatever the host shows (256), and more worker
    # processes than that would run on the other tenants' cores. Memory would allow more (2.5 GB per instance).
    workers = max(1, min(usable, 16))
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
                  f"in {tn:.1f} s (one single-threaded instance per core, like running libfri per image; capped at 16 = this one-GPU box's share of the "
                  f"{host_cores}-core host); 1 thread: {n1} image(s) in {t1:.1f} s",
    }


def extras(plan, ctx, torch, np, d_px, d_co, slots, alg_bytes, stream, full=False):
    """Event-timed figures for the kernels beside K1 and for the chain (see the call site). {us, frac}: microseconds per launch (or per chain /
    per image) and algorithmic bytes / us / 8 TB/s. Like the headline, every figure rotates over the bench's image / coefficient slots, so that its
    input comes from HBM (a kernel re-reading ONE 68 MB plane finds it in the 256 MiB Infinity Cache: K2 42 instead of 47 us, K3 26.5 instead of 29.6)."""
    import frave_amd

    F = plan.num_cells
    pstride, cstride = plan.pixel_bytes, plan.coef_count
    px0, co0 = d_px.data_ptr(), d_co.data_ptr()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=None):
        reps = reps or slots  # one pass over the slots (a second pass re-reads slot 0 only after 2 GB of other traffic)
        fn(slots - 1)
        torch.cuda.synchronize()
        ev0.record()
        for i in range(reps):
            fn(i % slots)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps * 1e3

    def entry(us, nbytes):
        return {"us": round(us, 2), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}

    px = lambda k: px0 + k * pstride
    co = lambda k: co0 + k * cstride * 4
    plane = F * 512
    k2_bytes = plane * (4 + 1 + 4) + 10 * 1024 * 4  # coefficient read + bucket + prediction write + histogram
    k4_bytes = plane * 4
    vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
    wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
    d_b = torch.empty(plane, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(plane, dtype=torch.int32, device="cuda")
    d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
    d_o = torch.empty(1, dtype=torch.int64, device="cuda")
    d_back = torch.empty(plan.pixel_bytes, dtype=torch.uint8, device="cuda")
    d_g = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
    d_w = torch.empty(18, dtype=torch.float64, device="cuda")
    d_par = torch.zeros(36, dtype=torch.float32, device="cuda")
    res = {"note": f"HIP events around {slots} back-to-back launches each, rotating over the {slots} image / coefficient slots (inputs from HBM), one 4096x4096 plane unless said "
                   "otherwise; frac = algorithmic bytes (SURVEY.md 8d) / us / 8 TB/s"}
    # K2 alone: the coefficients are K1's, so the plan may skip the exact-int32 guard launch (fri_hip_plan_assume_forward_coefficients)
    plan.assume_forward_coefficients(True)
    res["k2_predict_histogram"] = entry(timed(lambda k: plan.predict_histogram_dev(co(k), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=stream)), k2_bytes)
    plan.assume_forward_coefficients(False)
    assert int(d_h.sum()) + int(d_o) == plan.num_some
    res["k3_inverse"] = entry(timed(lambda k: plan.inverse_transform_dev(co(k), d_back.data_ptr(), stream=stream)), alg_bytes)
    assert torch.equal(d_back, d_px[(slots - 1) % slots]), "K3(K1(x)) != x"  # the last launch inverted slot slots - 1: the lossless round trip
    res["k4_fit_value_sums"] = entry(timed(lambda k: plan.fit_value_sums_dev(co(k), 0, d_g.data_ptr(), stream=stream)), k4_bytes)
    res["k4_fit_width_sums"] = entry(timed(lambda k: plan.fit_width_sums_dev(co(k), 0, vp, d_g.data_ptr(), d_w.data_ptr(), stream=stream)), k4_bytes)
    # the device part of FRIEncoder::encode for one image, everything in HBM, nothing but enqueues (fri_hip_encode_image_batch_dev): with the
    # parameters given (K1 -> K2) and with the fit (K1 -> value sums + solves -> width sums + solves -> K2: four launches, each sums kernel solves in its tail)
    d_par.copy_(torch.from_numpy(np.stack([vp, wp]).reshape(-1)))
    given = timed(lambda k: plan.encode_image_batch_dev(1, px(k), pstride, d_par.data_ptr(), co(k), cstride, d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(),
                                                        fit=False, stream=stream))
    res["chain_k1_k2_given_params"] = entry(given, alg_bytes + k2_bytes)
    fit = timed(lambda k: plan.encode_image_batch_dev(1, px(k), pstride, d_par.data_ptr(), co(k), cstride, d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(),
                                                      fit=True, stream=stream))
    res["chain_with_fit"] = entry(fit, alg_bytes + k2_bytes + 2 * k4_bytes)
    # ... and all the way to the emitter's input (fri_hip_encode_symbols_batch_dev): the scan writes one halfword per node instead of bucket + prediction,
    # the gather kernel (K5) puts them in the reference's stream order: 2 bytes per symbol is all that has to leave the device
    plan.set_stream_order()
    n_sym = plan.num_some
    d_words = torch.empty(plane, dtype=torch.uint16, device="cuda")
    d_sym = torch.empty(n_sym, dtype=torch.uint16, device="cuda")
    k2w_bytes = plane * (4 + 2) + 10 * 1024 * 4
    k5_bytes = n_sym * (4 + 2 + 2)  # order + halfword gathered + halfword written
    d_par.copy_(torch.from_numpy(np.stack([vp, wp]).reshape(-1)))
    sym = timed(lambda k: plan.encode_symbols_batch_dev(1, px(k), pstride, None, False, d_par.data_ptr(), co(k), cstride, d_words.data_ptr(), plane, d_sym.data_ptr(), n_sym,
                                                        d_h.data_ptr(), d_o.data_ptr(), stream=stream))
    res["chain_to_symbol_stream_given_params"] = entry(sym, alg_bytes + k2w_bytes + k5_bytes)
    res["k5_symbol_gather"] = entry(sym - given, k5_bytes)
    res["k5_symbol_gather"]["note"] = "difference of the two chains above (the scan's halfword form is ~1.5 us faster than its array form, so this slightly understates K5)"
    # The same chains with the fit, and with d_coefs = NULL: a caller that wants only the stream (the emitter does) lets the coefficients travel between the kernels as
    # the plan's compact planes - int16, None as 0 - instead of the ABI's int32 planes: K1 writes 34 instead of 68 MB, the fit and the scan read half (round 5; the
    # streams, histograms and parameters are the same bits: tests/test_gpu_compact.py). Algorithmic bytes stay SURVEY's int32 figures: the fractions compare like with like.
    d_rng = torch.zeros(1, dtype=torch.int64, device="cuda")
    symrun = lambda k, fit, coefs: plan.encode_symbols_batch_dev(1, px(k), pstride, None, fit, d_par.data_ptr(), coefs, cstride, d_words.data_ptr(), plane, d_sym.data_ptr(), n_sym,
                                                                d_h.data_ptr(), d_o.data_ptr(), d_rng.data_ptr(), stream=stream)
    res["chain_to_symbol_stream_with_fit"] = entry(timed(lambda k: symrun(k, True, co(k))), alg_bytes + k2w_bytes + k5_bytes + 2 * k4_bytes)
    d_par.copy_(torch.from_numpy(np.stack([vp, wp]).reshape(-1)))
    res["chain_to_symbol_stream_given_params_compact"] = entry(timed(lambda k: symrun(k, False, 0)), alg_bytes + k2w_bytes + k5_bytes)
    res["chain_to_symbol_stream_with_fit_compact"] = entry(timed(lambda k: symrun(k, True, 0)), alg_bytes + k2w_bytes + k5_bytes + 2 * k4_bytes)
    # ... and with d_node_words = NULL as well: the scan writes every symbol to its place in the stream (a table of stream positions, the inverse of the order); no gather kernel
    dirrun = lambda k, fit: plan.encode_symbols_batch_dev(1, px(k), pstride, None, fit, d_par.data_ptr(), 0, cstride, 0, plane, d_sym.data_ptr(), n_sym, d_h.data_ptr(), d_o.data_ptr(),
                                                          d_rng.data_ptr(), stream=stream)
    d_par.copy_(torch.from_numpy(np.stack([vp, wp]).reshape(-1)))
    res["chain_to_symbol_stream_given_params_direct"] = entry(timed(lambda k: dirrun(k, False)), alg_bytes + k2w_bytes + k5_bytes)
    res["chain_to_symbol_stream_with_fit_direct"] = entry(timed(lambda k: dirrun(k, True)), alg_bytes + k2w_bytes + k5_bytes + 2 * k4_bytes)
    for key in ("chain_to_symbol_stream_given_params_direct", "chain_to_symbol_stream_with_fit_direct"):
        res[key]["note"] = "d_coefs = d_node_words = NULL: int16 coefficient planes, the scan writes the stream itself (no node words, no gather kernel); same streams, histograms and parameters"
    for key in ("chain_to_symbol_stream_given_params_compact", "chain_to_symbol_stream_with_fit_compact"):
        res[key]["note"] = "d_coefs = NULL: int16 coefficient planes inside the chain (fri_hip_encode_symbols_batch_dev); same streams, histograms and parameters"
    del d_words, d_sym
    # K1 on RGB, the colour space libfri really encodes (wavelet_transform.rs:191, 415-416)
    plan3 = frave_amd.Plan(ctx, W, H, 3)
    try:
        tuned3 = plan3.tune_forward()  # like the headline's plan
    except frave_amd.api.FriHipError as e:
        tuned3 = {"winner": f"default (tuner: {e})"}
    n3 = 12  # rotating slots: 604 MB of pixels (6 slots: 302 MB, part of which still came out of the 256 MiB Infinity Cache: 52.8 against 53.7 us)
    d_px3 = torch.randint(0, 256, (n3, plan3.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co3 = torch.empty((n3, plan3.coef_count), dtype=torch.int32, device="cuda")
    # through the native loop like the headline (a Python call per launch leaves host gaps between 50-us kernels: 57.5-59.2 against 55.3 us in round 3)
    rgb = lambda n: plan3.time_transform_quant_dev(n3, d_px3.data_ptr(), plan3.pixel_bytes, d_co3.data_ptr(), plan3.coef_count, n, stream=stream)
    rgb(5 * n3)  # this plan's first launches: tables and buffers are cold
    res["k1_rgb"] = entry(rgb(10 * n3), plan3.pixel_bytes + plan3.coef_count * 4)
    res["k1_rgb"]["forward_tiling"] = tuned3.get("winner", "default")
    del d_px3, d_co3
    plan3.close()
    # K1 with many images per launch (the batch entry point; BASELINE config 4 runs like this), over the same slots: distinct images, every byte from and to HBM
    nb = min(slots, 32)
    usb = timed(lambda k: plan.transform_quant_dev(px0, co0, stream=stream, n_images=nb, pixel_stride=pstride, coef_stride=cstride), reps=4) / nb
    res["k1_batch_launch"] = {f"per_image_{nb}_distinct_images": entry(usb, alg_bytes), "note": f"one launch over {nb} distinct images ({nb * alg_bytes / 1e9:.1f} GB): HBM-bound like the headline"}
    if full:
        res["encode_pcie_bytes_per_image"] = {"host_to_device": plan.pixel_bytes, "device_to_host": plane * (4 + 1 + 4) + 10 * 1024 * 4 + 8}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)  # on top of the untimed spin-up, see SPIN_UP_LAUNCHES
    ap.add_argument("--slots", type=int, default=24, help="distinct image/coefficient buffer pairs the steps rotate over (>= 24: neither pixels nor coefficients can come from the 256 MiB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune", action="store_true", help="keep fri_hip_plan_create's default forward tiling instead of measuring (fri_hip_plan_tune_forward)")
    ap.add_argument("--no-extras", action="store_true", help="leave the event-timed figures of K2 / K3 / K4 / K1-RGB / the chain out of the line")
    ap.add_argument("--extras", action="store_true", help="(kept for older command lines: the extras are in the default line now; adds the PCIe byte counts)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU and never will - it starts the N ranks as a child
        # process (torch.distributed.run, one rank per GPU over RCCL), relays rank 0's line and exits with the child's code; a line for another
        # number of GPUs is an error, never a one-GPU number under an N-GPU command (frave_amd/dist.py, tests/test_bench_spawn.py).
        from frave_amd.dist import spawn_ranks

        code, line = spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        if line is not None and code == 0:
            print(json.dumps(line), flush=True)
        sys.exit(code)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the CPU leg runs first: it forks workers, which must happen before this process initialises the GPU
    cpu = cpu_baseline() if rank == 0 and world == 1 and not args.no_cpu_baseline else None

    import numpy as np
    import torch

    import frave_amd

    if world != args.gpus:  # a launcher started another number of ranks than the command asks for: refuse, do not measure something else
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("FRI_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path with a single rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:  # FRI_BENCH_FORCE_DIST=1 without a launcher: a one-rank job of its own
            from frave_amd.dist import _free_port

            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        # (RCCL prints a version banner on file descriptor 1 when the communicator is built - here and at the first barriers below; the contract is ONE JSON line
        # on stdout, so stdout points at stderr until the communicator exists)
        sys.stdout.flush()
        _saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = frave_amd.Context(local_rank)  # raises if the HIP library or a gfx950 GPU is missing: no fallback
    plan = frave_amd.Plan(ctx, W, H, CHANNELS)
    # The plan measures its forward tiling on THIS device (fri_hip_plan_tune_forward, like an FFT plan made with MEASURE): a handful of candidate cuts of the
    # cell lattice into tiles / shares, then the eight XCDs' shares balanced by their measured workgroup lifetimes - tens of milliseconds on scratch buffers of
    # the call's own, once, before anything is timed. Results do not depend on the tiling (every parity test runs on tuned and untuned plans alike).
    tuning = None
    if not args.no_tune:
        try:
            tuning = plan.tune_forward()
        except frave_amd.api.FriHipError as e:  # (e.g. no memory for the tuner's ~1 GB of scratch) - the plan keeps its default tiling and the line says so
            tuning = {"tuned": False, "error": str(e)}
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
        sys.stdout.flush()
        os.dup2(_saved_stdout, 1)  # the communicator exists: stdout is stdout again
        os.close(_saved_stdout)

    timed = {}

    def timed_steps():
        # K steps = K single-image launches over the rotating slots, issued by the library's native loop (what a C++ / Rust host
        # does through the same ABI; a Python call per step adds ~1 us of host time between launches). The same call brackets
        # the K launches with HIP events on the launch stream: the dominant kernel's mean launch period over the timed region.
        t_call = time.perf_counter()
        timed["kernel_us"] = plan.time_transform_quant_dev(args.slots, px0, pstride, co0, cstride, args.steps, stream=stream)
        timed["call_us"] = (time.perf_counter() - t_call) * 1e6

    # barrier + device synchronisation on both sides, MAX over ranks (the protocol tests/test_multi_gloo.py exercises on gloo)
    # (closing fence: the native call has polled its end event, so the launch stream - the only stream this rank has put work on - is idle when it
    # returns; hipStreamQuery proves it without the ~160 us a device-wide synchronise costs under an attached profiler. If the query says busy,
    # timed_region synchronises as before.)
    elapsed = timed_region(timed_steps, dist=dist, device_sync=torch.cuda.synchronize, device="cuda", device_idle=torch.cuda.current_stream().query)
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
            "forward_tiling": tuning if tuning is not None else "default (not measured: --no-tune)",
            "parallelism": f"batch sharded by image, image i -> GPU i mod {world} (fri_hip_shard_image), no collective",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(tiling=(tuning or {}).get("winner") or "interleaved/band16/cells8"),
            "kernel": "fwd_transform_quant_kernel<1,false,true,4,true,true,false,false>",
            "kernel_us": round(kernel_us, 3),
            "algorithmic_bytes_per_launch": alg_bytes,
        },
        # where the wall clock of the timed region goes (rank 0): the K launches by HIP events, what the native call adds before the first
        # and after the last of them, and the closing synchronise - a fixed 40-90 us that K = 20 feels and K = 400 does not
        "timed_region": {
            "wall_us": round(elapsed * 1e6, 1),
            "Mpixels_per_s_by_events": round(world * W * H / kernel_us, 1),  # the same steps by their HIP events (rank 0's kernel period): `value` is the wall-clock figure
            "kernels_us": round(kernel_us * args.steps, 1),
            "call_us": round(timed["call_us"], 1),
            "after_call_us": round(elapsed * 1e6 - timed["call_us"], 1),
        },
    }

    if rank == 0:
        # The batch form of the same kernel (fri_hip_transform_quant_batch_dev: BASELINE config 4 runs like this): `nb` DISTINCT images in one launch, per image.
        # One launch's ramp and tail are paid once per nb images: the steady state of the kernel. Its own PMC traffic pass: profiles/r05_k1_batch_traffic.json.
        nb = min(args.slots, 24)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        batch = lambda: plan.transform_quant_dev(px0, co0, stream=stream, n_images=nb, pixel_stride=pstride, coef_stride=cstride)
        batch()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(4):
            batch()
        ev1.record()
        torch.cuda.synchronize()
        us_img = ev0.elapsed_time(ev1) * 1e3 / 4 / nb
        out["roofline_batch"] = {
            "bound": "hbm", "images_per_launch": nb, "kernel_us_per_image": round(us_img, 3), "achieved": round(alg_bytes / us_img / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg_bytes / us_img / 1e3 / HBM_PEAK_GBS, 4), "traffic": measured_traffic("k1_batch_traffic", tiling=(tuning or {}).get("winner") or "interleaved/band16/cells8"),
            "note": f"one launch over {nb} distinct images (grid.y = {nb}), HIP events around 4 launches; traffic per image from the batch form's own PMC pass on this tiling (null: this tiling's batch form was not profiled)",
        }
        # ... and single-image launches dealt over two streams of the library (fri_hip_time_transform_quant_streams_dev): launch i + 1 fills the CUs launch i's
        # early finishers leave. A launch PERIOD (first begin to last end over the launches), never a kernel duration; never part of `value` or `roofline`.
        period = plan.time_transform_quant_streams_dev(args.slots, px0, pstride, co0, cstride, max(args.steps, 200), 2)
        out["two_stream_launch_period"] = {"us": round(period, 3), "frac_of_roofline_by_period": round(alg_bytes / period / 1e3 / HBM_PEAK_GBS, 4), "streams": 2,
                                          "note": "independent images, launches alternate over two streams; period, not kernel time"}

    # The rest of the path north_star names (K2 = predict + histogram) and of the encode chain, in the same line: every figure is the mean
    # period of back-to-back launches between two HIP events on the launch stream, with its fraction of the 8 TB/s roofline for the
    # algorithmic bytes of SURVEY.md section 8d. ~40 ms of GPU time behind the timed region (rank 0 only); never part of `value`.
    if rank == 0 and not args.no_extras:
        out["extras"] = extras(plan, ctx, torch, np, d_px, d_co, args.slots, alg_bytes, stream, full=args.extras)

    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        elif world > 1:
            out["cpu_baseline"] = None  # measured at N = 1 only (the contract: rank 0 at N = 1); the N = 1 line of the same round carries it
            out["cpu_baseline_note"] = "timed on rank 0 at N = 1 only; see the N = 1 line"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
