"""BASELINE config 4: 1024 x 4096x4096 images sharded per image over 8 MI355X, no collective.

The GPU test runs ONE GPU's share exactly as a rank of the 8-GPU job would (rank 0 of 8: images 0, 8, ..., 1016 = 128 images,
as 4 launches of 32 through fri_hip_transform_quant_batch_dev), with the partition function every multi-GPU path uses
(fri_hip_shard_size / fri_hip_shard_image). Sampled images are compared with the CPU oracle bit for bit, every image is checked by
the K3 round trip, and a batched image equals the same image in a launch of its own. The reference counterpart is the per-image
loop of crates/fri-cli/src/commands/bench.rs:15-120 around FRIEncoder::encode (encoder.rs:87-109).
"""
import numpy as np
import pytest

W = H = 4096
WORLD, N_TOTAL, PER_LAUNCH = 8, 1024, 32


def synthetic_image_dev(torch, global_index, n_bytes):
    """Image `global_index` of the synthetic batch, generated on the device: its content depends on the global index only."""
    g = torch.Generator(device="cuda").manual_seed(0xF7A5E000 + global_index)
    return torch.randint(0, 256, (n_bytes,), dtype=torch.uint8, device="cuda", generator=g)


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [0, 5])
def test_config4_one_gpus_share(oracle, rank):
    import torch

    import frave_amd as fa
    from frave_amd.dist import images_for_rank

    mine = images_for_rank(N_TOTAL, rank, WORLD)
    assert len(mine) == N_TOTAL // WORLD == 128 and all(i % WORLD == rank for i in mine)
    ctx = fa.Context(0)
    P = fa.Plan(ctx, W, H, 1)
    s = torch.cuda.current_stream().cuda_stream
    d_px = torch.empty((PER_LAUNCH, P.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.empty((PER_LAUNCH, P.coef_count), dtype=torch.int32, device="cuda")
    d_back = torch.empty(P.pixel_bytes, dtype=torch.uint8, device="cuda")
    d_single = torch.empty(P.coef_count, dtype=torch.int32, device="cuda")
    sampled = {mine[0], mine[37], mine[-1]} if rank == 0 else {mine[64]}
    checked = 0
    for launch in range(len(mine) // PER_LAUNCH):
        ids = mine[launch * PER_LAUNCH:(launch + 1) * PER_LAUNCH]
        for k, gi in enumerate(ids):
            d_px[k].copy_(synthetic_image_dev(torch, gi, P.pixel_bytes))
        P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=PER_LAUNCH, pixel_stride=P.pixel_bytes, coef_stride=P.coef_count)
        for k, gi in enumerate(ids):
            # every image: lossless through the inverse kernel
            P.inverse_transform_dev(d_co[k].data_ptr(), d_back.data_ptr(), stream=s)
            assert torch.equal(d_back, d_px[k]), f"image {gi}: K3(K1(x)) != x"
            if gi in sampled:
                img = d_px[k].cpu().numpy()
                Wv = oracle.Wavelet(img, H, W, 1)
                assert np.array_equal(d_co[k].cpu().numpy().reshape(1, P.num_cells, 512), Wv.coefficients()), f"image {gi} differs from the oracle"
                Wv.close()
                # and the batched launch gives what a launch of its own gives
                P.transform_quant_dev(d_px[k].data_ptr(), d_single.data_ptr(), stream=s)
                assert torch.equal(d_single, d_co[k])
                checked += 1
    assert checked == len(sampled)
    P.close()
    ctx.close()


@pytest.mark.gpu
def test_multi_device_helper_on_the_devices_present(oracle):
    """fri_hip_multi: one host thread + ctx + plan per device; with one GPU it degenerates to the host batch path."""
    import torch

    import frave_amd as fa
    from tests.common import gen_image

    n_dev = min(torch.cuda.device_count(), 2)
    w, h, c = 640, 360, 3
    M = fa.Multi(list(range(n_dev)), w, h, c)
    imgs = [gen_image("noise" if i % 2 else "smooth", w, h, c, i) for i in range(7)]
    outs = M.transform_quant(imgs)
    for i in (0, 3, 6):
        Wv = oracle.Wavelet(imgs[i], h, w, c)
        assert np.array_equal(outs[i], Wv.coefficients())
        Wv.close()
    M.close()


@pytest.mark.gpu
def test_sharded_encode_on_the_devices_present(oracle):
    """fri_hip_multi_encode_image: the reference's per-image loop (bench.rs:15-120 around FRIEncoder::encode) over the GPUs of the node - image i on
    device i mod N, one host thread per device, each image the whole asynchronous chain (K1 -> fit -> K2) with overlapped copies. Sampled images
    against the oracle run with the parameters the chain fitted; every image's histogram total; given parameters (fit = 0) as well."""
    import torch

    import frave_amd as fa
    from tests.common import gen_image, random_params

    n_dev = min(torch.cuda.device_count(), 2)
    w, h, c = 640, 360, 3
    M = fa.Multi(list(range(n_dev)), w, h, c)
    imgs = [gen_image("noise" if i % 2 else "smooth", w, h, c, 30 + i) for i in range(7)]
    coefs, par, bucket, pred, hist, oob = M.encode_image(imgs, fit=True)
    L = fa.load_library()
    some = L.fri_hip_plan_num_some(L.fri_hip_multi_plan(M._h, 0))
    for i in range(7):
        assert all(int(hist[i][ch].sum()) + int(oob[i][ch]) == some for ch in range(c)), i
    for i in (0, 3, 6):
        Wv = oracle.Wavelet(imgs[i], h, w, c)
        assert np.array_equal(coefs[i], Wv.coefficients())
        Wv.quantize(np.ones(32, np.int32))
        for ch in range(c):
            wb, wpred, whist, woob = Wv.predict(ch, par[i][ch, 0], par[i][ch, 1])
            assert np.array_equal(bucket[i][ch], wb) and np.array_equal(pred[i][ch], wpred) and np.array_equal(hist[i][ch], whist) and int(oob[i][ch]) == woob
        Wv.close()
    given = [np.stack([np.stack(random_params(50 + i + ch)) for ch in range(c)]) for i in range(7)]
    coefs2, par2, bucket2, pred2, hist2, oob2 = M.encode_image(imgs, fit=False, params=given, want_bucket=False)
    assert bucket2 is None and all(np.array_equal(a, b) for a, b in zip(par2, [g.astype(np.float32) for g in given]))
    for i in (1, 4):
        Wv = oracle.Wavelet(imgs[i], h, w, c)
        Wv.quantize(np.ones(32, np.int32))
        for ch in range(c):
            wb, wpred, whist, woob = Wv.predict(ch, given[i][ch, 0], given[i][ch, 1])
            assert np.array_equal(pred2[i][ch], wpred) and np.array_equal(hist2[i][ch], whist) and int(oob2[i][ch]) == woob
        Wv.close()
    M.close()


def test_shard_functions_cover_the_batch_once():
    """The C partition itself (host-only: no GPU): disjoint, complete, balanced; image i -> shard i mod n."""
    import frave_amd as fa

    L = fa.load_library()
    for n_images, n_shards in [(1024, 8), (7, 2), (3, 8), (0, 4), (1000, 3)]:
        seen = []
        for sh in range(n_shards):
            size = L.fri_hip_shard_size(n_images, sh, n_shards)
            ids = [L.fri_hip_shard_image(k, sh, n_shards) for k in range(size)]
            assert all(i % n_shards == sh and i < n_images for i in ids)
            seen += ids
        assert sorted(seen) == list(range(n_images))
    assert L.fri_hip_shard_size(10, 4, 4) == 0 and L.fri_hip_shard_size(10, 0, 0) == 0


@pytest.mark.gpu
def test_two_device_threads_on_one_gpu(oracle):
    """Multi([0, 0], ...): the same device twice. A one-GPU box cannot run image i -> GPU i mod N on N > 1 GPUs, but it can run everything else the N > 1 path is made
    of: two host threads, two contexts, two plans, two sets of streams, pinned pools and accumulators working CONCURRENTLY (against one GPU), the partition
    i mod 2 dealing the images to them. Every image of both halves against the oracle - coefficients for the transform path; buckets, predictions and
    histograms (with the parameters the chain fitted) for the sharded encode."""
    import frave_amd as fa
    from tests.common import gen_image

    w, h, c = 640, 360, 3
    M = fa.Multi([0, 0], w, h, c)
    assert M.num_devices == 2
    imgs = [gen_image("noise" if i % 3 else "smooth", w, h, c, 70 + i) for i in range(10)]
    for rep in range(2):  # (the second pass reuses the pinned pools and streams of the first)
        outs = M.transform_quant(imgs)
        coefs, par, bucket, pred, hist, oob = M.encode_image(imgs, fit=True)
    L = fa.load_library()
    some = L.fri_hip_plan_num_some(L.fri_hip_multi_plan(M._h, 1))
    for i in range(len(imgs)):
        Wv = oracle.Wavelet(imgs[i], h, w, c)
        assert np.array_equal(outs[i], Wv.coefficients()), i
        assert np.array_equal(coefs[i], Wv.coefficients()), i
        Wv.quantize(np.ones(32, np.int32))
        for ch in range(c):
            wb, wpred, whist, woob = Wv.predict(ch, par[i][ch, 0], par[i][ch, 1])
            assert int(hist[i][ch].sum()) + int(oob[i][ch]) == some
            assert np.array_equal(bucket[i][ch], wb) and np.array_equal(pred[i][ch], wpred) and np.array_equal(hist[i][ch], whist) and int(oob[i][ch]) == woob, (i, ch)
        Wv.close()
    M.close()


@pytest.mark.gpu
def test_driver_batch_to_frv_bytes():
    """fri_driver batch-frv: n images -> n .frv byte strings with the device chains (to the emitter's input) and the host rANS emits pipelined
    (libfri::encode_batch_bytes) - one device thread, and two device threads on the one GPU (--same-device: the N > 1 threading without N GPUs). The driver
    itself checks that equal inputs give equal bytes, that image 0 equals FRIEncoder::encode_bytes_streamed's output and that it decodes to its input."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = os.path.join(root, "frave_amd", "host", "fri_driver")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "frave_amd", "host")])
    for extra in ([], ["--gpus", "2", "--same-device"]):
        out = subprocess.run([driver, "batch-frv", "640", "360", "3", "12", "--emitters", "3"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr + out.stdout
        assert "decodes losslessly" in out.stdout
