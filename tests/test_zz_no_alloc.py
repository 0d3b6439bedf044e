"""mipt_tonemap_device_noalloc (the entry point that allocates nothing: Reinhard's partial sums live in a file-scope
device array) against mipt_tonemap_device, through the C ABI: bit-identical output on two calls in a row, the second
on a smaller image, so that whatever the first call left in the shared array is still there when the second runs.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from pathtracer_amd import api, capi

pytestmark = pytest.mark.gpu


def _hdr_image(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=np.float32) ** 3 * 8.0
    img[rng.random((h, w)) < 0.01] *= 50.0   # a few very bright pixels, as emitters leave them
    return img.astype(np.float32)


def _read(ctx, tensor):
    out = np.zeros(tuple(tensor.shape), np.float32)
    capi.check(capi.dll().mipt_device_floats_read(ctx.h, C.c_void_p(tensor.data_ptr()), C.c_float(1.0), out.size, capi.fp(out)),
               "mipt_device_floats_read")   # runs on the context's stream, after the kernels queued before it
    return out


@pytest.mark.parametrize("kind", [0, 1])
def test_noalloc_tonemap_equals_tonemap_device_on_two_calls_in_a_row(kind):
    ctx = api.Context()
    try:
        for (h, w, seed) in [(270, 480, 1), (61, 97, 2)]:   # the second image is smaller (fewer reduction blocks)
            img = _hdr_image(h, w, seed)
            d_in = torch.from_numpy(img).cuda()
            d_want = torch.full_like(d_in, -1.0)
            d_got = torch.full_like(d_in, -2.0)
            torch.cuda.synchronize()
            ctx.tonemap_device(kind, d_in.data_ptr(), d_want.data_ptr(), w, h)
            capi.check(capi.dll().mipt_tonemap_device_noalloc(ctx.h, kind, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_got.data_ptr()), w, h),
                       "mipt_tonemap_device_noalloc")
            got, want = _read(ctx, d_got), _read(ctx, d_want)
            assert np.isfinite(want).all() and want.min() >= 0.0 and want.max() <= 1.0 and want.std() > 0.01
            assert np.array_equal(got, want), f"kind {kind}, {w}x{h}: max |d| = {np.abs(got - want).max()}"
    finally:
        ctx.close()


def test_noalloc_tonemap_refuses_what_it_cannot_take():
    ctx = api.Context()
    try:
        d = torch.zeros(12, device="cuda")
        f = capi.dll().mipt_tonemap_device_noalloc
        assert f(ctx.h, 0, C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), (1 << 26) + 1, 1) == -6   # MIPT_ERR_TOO_LARGE
        assert f(ctx.h, 0, C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 1 << 14, (1 << 12) + 1) == -6
        assert f(ctx.h, 2, C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 2, 2) == -1            # MIPT_ERR_ARGUMENT
        assert f(ctx.h, 0, None, C.c_void_p(d.data_ptr()), 2, 2) == -1
    finally:
        ctx.close()
