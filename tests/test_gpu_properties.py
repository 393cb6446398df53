"""GPU (-m gpu): size-independent properties at BASELINE's full batch size (config 3: 240 frames of
3840x2160 12-bit): every frame round-trips, the result does not depend on the order or the
neighbours of a frame in the batch, and decoding twice gives the same bytes (checksum of checksums)."""
import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.gpu

W, H, N, DISTINCT = 3840, 2160, 240, 12


def _sums(torch, t_out, n):
    v = t_out.view(torch.int32).view(n, -1)
    # two independent folds per frame: plain sum and position-weighted sum (wrapping int64)
    idx = torch.arange(v.shape[1], device=v.device, dtype=torch.int64) % 65521 + 1
    return (v.to(torch.int64).sum(1).cpu().numpy(), (v.to(torch.int64) * idx).sum(1).cpu().numpy())


def test_full_batch_order_and_repeat_invariance(gpu_ctx):
    import torch
    import motioncam_decoder_amd as M
    dev = torch.device("cuda:0")
    imgs = [L.synth_image(W, H, 12, i % 2, 12.0, 3000 + i) for i in range(DISTINCT)]  # U and Nat alternate
    bufs = [L.encode7(im) for im in imgs]
    t_bufs = [torch.from_numpy(b).to(dev) for b in bufs]
    t_out = torch.zeros(N * W * H * 2, dtype=torch.uint8, device=dev)

    def run(order):
        t_out.zero_()
        descs = [(t_bufs[k].data_ptr(), t_bufs[k].numel(), W, H, 7, t_out.data_ptr() + i * W * H * 2, W * H)
                 for i, k in enumerate(order)]
        torch.cuda.synchronize()
        written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
        assert status == [0] * N and written == [W * H] * N
        return _sums(torch, t_out, N)

    rng = np.random.default_rng(3)
    order_a = [i % DISTINCT for i in range(N)]
    order_b = list(rng.permutation(order_a))
    a1 = run(order_a)
    a2 = run(order_a)
    b = run(order_b)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])       # deterministic
    # the expected folds of each distinct frame, from the source images (round trip)
    ref = {}
    for k, im in enumerate(imgs):
        v = torch.from_numpy(im.view(np.int32).reshape(1, -1).copy()).to(dev).view(torch.uint8)
        s = _sums(torch, v, 1)
        ref[k] = (int(s[0][0]), int(s[1][0]))
    for i, k in enumerate(order_a):
        assert (int(a1[0][i]), int(a1[1][i])) == ref[k], (i, k)
    for i, k in enumerate(order_b):                                             # order / neighbours do not matter
        assert (int(b[0][i]), int(b[1][i])) == ref[int(k)], (i, k)


def test_frame_checksums_do_not_depend_on_the_gpu_count(gpu_ctx):
    """BASELINE config 5: 7680x4320 frames sharded over 1, 2, 4, 8 ranks (frame i -> rank i mod world,
    motioncam_decoder_amd/shard.py); the per-frame checksums of the job must be the same for every world
    size.  One GPU plays the ranks one after another."""
    import torch
    import motioncam_decoder_amd as M
    from motioncam_decoder_amd import shard
    dev = torch.device("cuda:0")
    w, h, total, distinct = 7680, 4320, 16, 4
    imgs = [L.synth_image(w, h, 12, 1, 12.0, 5000 + i) for i in range(distinct)]
    bufs = [torch.from_numpy(L.encode7(im)).to(dev) for im in imgs]
    want = [shard.frame_checksum(imgs[i % distinct]) for i in range(total)]
    for world in (1, 2, 4, 8):
        got = {}
        for rank in range(world):
            mine = shard.shard_frames(total, rank, world)
            t_out = torch.zeros(len(mine) * w * h * 2, dtype=torch.uint8, device=dev)
            descs = [(bufs[g % distinct].data_ptr(), bufs[g % distinct].numel(), w, h, 7,
                      t_out.data_ptr() + k * w * h * 2, w * h) for k, g in enumerate(mine)]
            written, status = gpu_ctx.decode_batch(M.Context.make_frames(descs))
            assert status == [0] * len(mine) and written == [w * h] * len(mine)
            out = t_out.cpu().numpy().view(np.uint16).reshape(len(mine), h, w)
            for k, g in enumerate(mine):
                got[g] = shard.frame_checksum(out[k])
            del t_out
        assert [got[i] for i in range(total)] == want, world
