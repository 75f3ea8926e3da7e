"""rt_bloom_static (the entry point that allocates nothing) against rt_bloom, through the C ABI, bit for bit: two calls
in a row on one GPU, the second with a smaller frame, so that the static scratch array still holds the first call's
rows when the second runs.  Also checked: the result equals the CPU oracle's and a frame beyond the array is rejected."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def _frame(H, W, seed):
    rng = np.random.RandomState(seed)
    img = (rng.uniform(0, 1.4, (H, W, 3)) ** 3).astype(np.float32)
    img[H // 4:H // 4 + 10, W // 3:W // 3 + 40] = 3.0
    return img


def _bloom_both(torch, img, F, thr, inten):
    H, W = img.shape[:2]
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.from_numpy(img).cuda()
    scratch = torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device="cuda")
    api.bloom_device(a.data_ptr(), scratch.data_ptr(), W, H, F, thr, inten, stream=stream)
    b = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    api.bloom_static(b.data_ptr(), W, H, F, thr, inten, stream=stream)  # allocates nothing: inputs and outputs are the caller's
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_bloom_equals_workspace_bloom_twice():
    torch = pytest.importorskip("torch")
    # first call: a frame as large as the headline configuration's; second call: a smaller, ragged one
    for (H, W, F, thr, inten, seed) in ((1080, 1920, 15, 0.5, 0.8, 1), (203, 333, 30, 0.3, 1.1, 2)):
        img = _frame(H, W, seed)
        want, got = _bloom_both(torch, img, F, thr, inten)
        bad = np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))
        print(f"rt_bloom_static {W}x{H} F={F}: {bad} words differ from rt_bloom")
        assert bad == 0
        assert np.abs(want - img).max() > 0.01  # the bloom did something
        if H * W <= 203 * 333:  # the oracle's serial loops: small frame only
            ref = O.bloom(img, F, thr, inten)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_static_bloom_rejects_frames_beyond_its_array():
    torch = pytest.importorskip("torch")
    lib = api.lib()
    d = torch.zeros(16, dtype=torch.float32, device="cuda")
    W, H = 2048, 1025  # one row more than the array holds
    assert W * H > api.RT_BLOOM_STATIC_MAX_PIXELS
    rc = lib.rt_bloom_static(d.data_ptr(), W, H, 15, C.c_float(0.5), C.c_float(0.8), None)
    assert rc == api.RT_ERR_LIMIT
    assert b"rt_bloom_static" in lib.rt_last_error()
    assert lib.rt_bloom_static(None, 8, 8, 1, C.c_float(0.5), C.c_float(0.8), None) == api.RT_ERR_INVALID
    # the largest frame it accepts
    W, H = 2048, 1024
    img = _frame(H, W, 3)
    want, got = _bloom_both(torch, img, 1, 0.5, 0.8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
