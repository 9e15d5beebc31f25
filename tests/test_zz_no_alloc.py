"""GPU: mj_decode_static, the entry point that allocates no device memory, against
mj_batch_create + mj_batch_decode on the same files: two calls in a row, the second with
a smaller input; coefficients and RGB bit-exact."""
import numpy as np
import pytest

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import load_fixture

pytestmark = pytest.mark.gpu


def _expected(ctx, files, flags):
    b = ctx.batch(files, flags)
    b.decode()
    out = [(b.coefs(i), b.rgb(i), b.info(i)) for i in range(len(files))]
    b.close()
    return out


def _run_static(ctx, files, flags, coefs_t, rgb_t):
    import torch
    view = ctx.decode_static(files, coefs_t.data_ptr(), coefs_t.numel() * 2, rgb_t.data_ptr(), rgb_t.numel(), flags)
    torch.cuda.synchronize()
    got = []
    bpp = 4 if flags & binding.MJ_OUT_RGBA else 3
    for i in range(len(files)):
        inf = view.info(i)
        # the caller's buffer holds blocks in natural order; compare in zigzag order, as mj_batch_copy_coefs reports them
        c = coefs_t[inf.coef_offset * 64:(inf.coef_offset + inf.n_blocks) * 64].cpu().numpy().reshape(-1, 64)[:, binding.NATPOS]
        raw = rgb_t[inf.rgb_offset:inf.rgb_offset + inf.rgb_pitch * inf.Y].cpu().numpy().reshape(inf.Y, inf.rgb_pitch)
        got.append((c, raw[:, :bpp * inf.X].reshape(inf.Y, inf.X, bpp), view.status(i)))
    return got


@pytest.mark.parametrize("flags", [0, binding.MJ_OUT_RGBA])
def test_static_entry_equals_allocating_entry_twice(flags):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    ctx = J.Context(0)
    big = [J.synth.config_jpeg("C3", seed=3), load_fixture("c422_q95_opt_136x40_ri5")[0], J.synth.config_jpeg("C1", seed=1),
           load_fixture("g_q75_72x40_ri4")[0], J.synth.config_jpeg("C2", seed=5)]
    small = [load_fixture("c420_q85_100x75_ri3")[0], J.synth.encode_jpeg(J.synth.synth_image(200, 120, 9), "4:4:4", 80, 0)]
    coefs_t = torch.full((40 << 20,), 0x5A5A, dtype=torch.int16, device="cuda:0")      # 80 MiB, stale contents
    rgb_t = torch.full((64 << 20,), 0xA5, dtype=torch.uint8, device="cuda:0")
    free_after_first = None
    for files in (big, small, big):
        want = _expected(ctx, files, flags)
        got = _run_static(ctx, files, flags, coefs_t, rgb_t)
        if free_after_first is not None:
            # the call allocates no device memory: after the first call (which may still make the
            # runtime load code) the free memory of the device is the same before and after
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(0)[0]
            view = ctx.decode_static(files, coefs_t.data_ptr(), coefs_t.numel() * 2, rgb_t.data_ptr(), rgb_t.numel(), flags)
            view.sync()
            assert torch.cuda.mem_get_info(0)[0] == free0, "mj_decode_static changed the device's free memory"
        free_after_first = True
        for i, ((wc, wr, _), (gc, gr, st)) in enumerate(zip(want, got)):
            assert st == 0, i
            assert np.array_equal(gc, wc), f"image {i}: coefficients differ"
            assert np.array_equal(gr, wr), f"image {i}: RGB differs"
    ctx.close()


def test_static_entry_refuses_what_does_not_fit():
    import torch
    ctx = J.Context(0)
    files = [J.synth.config_jpeg("C3", seed=s) for s in range(16)]      # ~16 x 6.5 MB of work memory > 64 MiB
    t = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(RuntimeError, match="static limits"):
        ctx.decode_static(files, t.data_ptr(), t.numel(), t.data_ptr(), t.numel(), 0)
    one = [load_fixture("c420_q85_64x48")[0]]
    with pytest.raises(RuntimeError, match="outputs need"):
        ctx.decode_static(one, t.data_ptr(), 64, t.data_ptr(), 64, 0)
    with pytest.raises(RuntimeError, match="only MJ_OUT_RGBA"):
        ctx.decode_static(one, t.data_ptr(), t.numel(), t.data_ptr(), t.numel(), binding.MJ_WANT_PLANES)
    scans = [J.synth.encode_jpeg(J.synth.synth_image(64, 48, 2), "4:2:0", 85, 0, scans_per_component=True)]
    with pytest.raises(RuntimeError, match="several scans"):
        ctx.decode_static(scans, t.data_ptr(), t.numel(), t.data_ptr(), t.numel(), 0)
    ctx.close()
