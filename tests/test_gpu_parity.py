"""GPU parity: the HIP path, called through the C ABI, against (a) the committed
reference outputs in tests/golden and (b) the CPU restatement in oracle/ on seeded
synthetic inputs.  Bar: coefficients, planes and RGB bit-exact (the north star asks
for +-1 on RGB; the kernels reproduce the reference's rounding points, so the tests
ask for 0 and would report the max error)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import (PARITY_FIXTURES, SCANS_PER_COMPONENT_CASES, load_fixture, scans_per_component_pair, wrap_triple_pixels,
                      ycc_neighbourhood_jpeg)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    c = J.Context(0)
    yield c
    c.close()


def _check_image(b, i, coefs, planes, rgb, what):
    assert b.status(i) == 0, f"{what}: status {b.status(i)}"
    got = b.coefs(i)
    assert got.shape == coefs.shape, what
    bad = np.nonzero((got != coefs).any(axis=1))[0]
    assert bad.size == 0, f"{what}: {bad.size} coefficient blocks differ, first {bad[:5]}"
    if planes is not None:
        for c, p in enumerate(planes):
            assert np.array_equal(b.plane(i, c), p), f"{what}: plane {c} differs"
    out = b.rgb(i)
    err = np.abs(out.astype(int) - rgb.astype(int)).max() if out.size else 0
    assert err == 0, f"{what}: RGB max abs error {err}"


@pytest.mark.parametrize("flags", [0, binding.MJ_ENTROPY_SERIAL])
def test_golden_fixtures_one_mixed_batch(ctx, flags):
    """All fixtures (grey, 4:2:0, 4:2:2, 4:4:4; DRI on/off; optimised tables;
    custom quantisers) in ONE batch, against the reference's committed outputs."""
    files, golds = [], []
    for name in PARITY_FIXTURES:
        d, g = load_fixture(name)
        files.append(d)
        golds.append(g)
    b = ctx.batch(files, binding.MJ_WANT_PLANES | flags)
    b.decode()
    b.sync()
    for i, (name, g) in enumerate(zip(PARITY_FIXTURES, golds)):
        planes = [g[f"plane{c}"] for c in range(3) if f"plane{c}" in g]
        _check_image(b, i, g["coefs"], planes, g["rgb"], name)
    b.close()


@pytest.mark.parametrize("name", PARITY_FIXTURES)
def test_golden_fixture_alone_and_twice(ctx, name):
    """Each fixture as a batch of one, decoded twice (idempotence of a resident batch)."""
    d, g = load_fixture(name)
    b = ctx.batch([d], 0)
    for _ in range(2):
        b.decode()
        b.sync()
        _check_image(b, 0, g["coefs"], None, g["rgb"], name)
    b.close()


def test_last_mcu_divergence_is_decoded_by_count(ctx, oracle):
    """The reference drops a final MCU that starts in the last entropy byte
    (decode.c:578-589); the GPU path decodes by MCU count like the restatement."""
    d, g = load_fixture("g_flat_32x8_refdrops")
    b = ctx.batch([d], 0)
    b.decode()
    o = oracle.decode(d)
    assert np.array_equal(b.coefs(0), o.coefs)
    assert np.array_equal(b.coefs(0)[:3], g["coefs"])
    assert np.array_equal(b.rgb(0), o.rgb)
    b.close()


def test_file_the_reference_dies_on_is_decoded_whole(ctx, oracle):
    """A partly consumed data FF before a restart marker (FF 00 FF Dn) sends the reference's
    cursor onto the marker and the process to its end (bitstream.c:84-151).  Here markers are
    found by the byte classifier: the whole file decodes, equal to the restatement, and the
    blocks the reference tapped before it died are a prefix of it."""
    from conftest import FATAL_FIXTURES
    assert FATAL_FIXTURES
    for name in FATAL_FIXTURES:
        d, g = load_fixture(name)
        b = ctx.batch([d], 0)
        b.decode()
        o = oracle.decode(d)
        assert b.status(0) == 0
        assert np.array_equal(b.coefs(0), o.coefs)
        assert np.array_equal(b.coefs(0)[:g["coefs"].shape[0]], g["coefs"])
        assert np.array_equal(b.rgb(0), o.rgb)
        b.close()


SYNTH_CASES = [
    # (w, h, sampling, quality, restart interval in MCUs, grey)
    (200, 120, "4:2:0", 85, 0, False),
    (200, 120, "4:2:0", 50, 13, False),      # intervals that straddle MCU rows
    (333, 77, "4:2:2", 95, 21, False),       # width/height not MCU multiples
    (131, 260, "4:4:4", 75, 17, False),
    (96, 200, "4:4:0", 85, 12, False),       # 1x2 sampling (not producible by the fixture encoder)
    (256, 64, "4:1:1", 85, 8, False),        # 4x1 sampling
    (640, 480, None, 50, 0, True),           # grey, BASELINE config 5
    (640, 480, "4:2:2", 95, 0, False),
    (640, 480, "4:2:0", 75, 40, False),
    (48, 16, "4:2:0", 90, 1, False),         # one MCU per interval
]


@pytest.mark.parametrize("case", SYNTH_CASES, ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}_q{c[3]}_ri{c[4]}")
def test_synthetic_vs_oracle(ctx, oracle, case):
    w, h, samp, q, ri, grey = case
    img = J.synth.synth_image(w, h, seed=w * 7 + h, grey=grey)
    data = J.synth.encode_jpeg(img, samp or "4:4:4", q, ri)
    o = oracle.decode(data)
    b = ctx.batch([data], binding.MJ_WANT_PLANES)
    b.decode()
    _check_image(b, 0, o.coefs, o.planes, o.rgb, str(case))
    b.close()


@pytest.mark.parametrize("sampling", ["4:4:4", "4:2:0"])
@pytest.mark.parametrize("rgba", [0, binding.MJ_OUT_RGBA])
def test_green_that_wraps_to_zero_like_the_reference(ctx, oracle, sampling, rgba):
    """Flat MCUs around (Y, Cb, Cr) = (219, 228, 28): the one triple whose green wraps to 0
    in the reference (colour_conversion.c:71-79), with its clamped neighbours, on the
    vector path (interior) of the colour step."""
    data = ycc_neighbourhood_jpeg(sampling)
    o = oracle.decode(data)
    hit = wrap_triple_pixels(o.planes, sampling)
    assert hit.sum() >= 64, "the file no longer holds the wrapping triple"
    assert set(o.rgb[hit][:, 1].tolist()) == {0}
    b = ctx.batch([data], binding.MJ_WANT_PLANES | rgba)
    b.decode()
    assert b.status(0) == 0
    for c, p in enumerate(o.planes):
        assert np.array_equal(b.plane(0, c), p), f"plane {c}"
    out = b.rgb(0)
    assert np.array_equal(out[..., :3], o.rgb)
    if rgba:
        assert (out[..., 3] == 255).all()
    b.close()


@pytest.mark.parametrize("flags", [0, binding.MJ_ENTROPY_SERIAL])
def test_tables_that_need_the_canonical_walk(ctx, oracle, flags):
    """Files re-coded with AC tables whose frequent symbols have 11-bit codes over 76 ten-bit
    prefixes (tests/huffcraft.py): most symbols are found by the canonical ranges
    (decode.c:674-681), in the lane-parallel passes, the wavefront-wide lone decoder, the
    cross-sequence passes (the scans have no restart markers) and the writing pass.  One
    batch of a 640 x 480 4:2:0, a grey and a 4:4:4 file."""
    import huffcraft
    files, want = [], []
    for w, h, sampling, pattern, grey in ((640, 480, "4:2:0", [0, 0, 0, 0, 1, 2], False), (320, 200, "4:4:4", [0], True),
                                          (200, 120, "4:4:4", [0, 1, 2], False)):
        src = J.synth.encode_jpeg(J.synth.synth_image(w, h, seed=w + h, grey=grey), sampling, 90, 0)
        o = oracle.decode(src)
        data, tabs = huffcraft.recode(src, o.coefs, pattern)
        assert all(huffcraft.long_prefixes(t) > 8 for t in tabs.values())
        files.append(data)
        want.append(oracle.decode(data))
        assert np.array_equal(want[-1].coefs, o.coefs)
    b = ctx.batch(files, binding.MJ_WANT_PLANES | flags)
    b.decode()
    for i, o in enumerate(want):
        _check_image(b, i, o.coefs, o.planes, o.rgb, f"re-coded file {i}")
    b.close()


def test_tables_with_long_dc_codes_and_sixteen_bit_ac_codes(ctx, oracle):
    """DC differences coded with 12-bit codes and nearly all AC symbols with 16-bit codes
    (tests/huffcraft.py): the second-level lookups of every pass, DC symbols included, on a
    scan long enough for the cross-sequence passes."""
    import huffcraft
    src = J.synth.encode_jpeg(J.synth.synth_image(480, 320, seed=11), "4:2:0", 90, 0)
    o = oracle.decode(src)
    data, _ = huffcraft.recode(src, o.coefs, [0, 0, 0, 0, 1, 2], huffcraft.sixteen_bit_ac_table, {2: [0], 12: list(range(1, 12))})
    want = oracle.decode(data)
    assert np.array_equal(want.coefs, o.coefs)
    for flags in (0, binding.MJ_ENTROPY_SERIAL):
        b = ctx.batch([data, src], binding.MJ_WANT_PLANES | flags)
        b.decode()
        _check_image(b, 0, want.coefs, want.planes, want.rgb, "re-coded file")
        _check_image(b, 1, o.coefs, o.planes, o.rgb, "its source")
        b.close()


@pytest.mark.parametrize("flags", [0, binding.MJ_OUT_RGBA, binding.MJ_ENTROPY_SERIAL])
def test_frames_coded_in_one_scan_per_component(ctx, oracle, flags):
    """Files with three scans of one component each (T.81 A.2.2), mixed with their interleaved
    twins in one batch: coefficients (scan after scan), the planes of the components and RGB
    against the restatement, and RGB against the twin (which the reference decodes).  Parity
    with the reference itself is unpinned for these files: it has no output for them."""
    files, want = [], []
    for case in SCANS_PER_COMPONENT_CASES:
        multi, inter = scans_per_component_pair(case)
        files += [multi, inter]
        want += [oracle.decode(multi), oracle.decode(inter)]
    big = J.synth.encode_jpeg(J.synth.synth_image(1280, 720, seed=4), "4:2:0", 88, 0, scans_per_component=True)
    files.append(big)
    want.append(oracle.decode(big))
    b = ctx.batch(files, flags)
    assert len(files) == C.c_int(b.lib.mj_batch_size(b.h)).value
    b.decode()
    for i, o in enumerate(want):
        assert b.status(i) == 0, i
        assert np.array_equal(b.coefs(i), o.coefs), f"file {i}: coefficients"
        out = b.rgb(i)
        assert np.array_equal(out[..., :3], o.rgb), f"file {i}: RGB"
        if flags & binding.MJ_OUT_RGBA:
            assert (out[..., 3] == 255).all()
        if i % 2 == 0:
            for c, p in enumerate(o.planes):
                assert np.array_equal(b.plane(i, c), p), f"file {i}: plane {c}"
            if i + 1 < len(want) and i < 2 * len(SCANS_PER_COMPONENT_CASES):
                assert np.array_equal(o.rgb, want[i + 1].rgb)
    b.close()


@pytest.mark.parametrize("flags", [0, binding.MJ_OUT_RGBA, binding.MJ_ENTROPY_SERIAL])
def test_frames_coded_in_scans_of_some_of_the_components(ctx, oracle, flags):
    """Files of two scans -- luma alone then both chroma components interleaved, or luma and Cb
    interleaved then Cr alone (T.81 A.2.2, A.2.3) -- mixed with their interleaved twins and with
    files of one scan per component in one batch: coefficients (scan after scan), the planes
    and RGB against the restatement, and RGB against the twin (which the reference decodes).
    Parity with the reference itself is unpinned: it has no output for files of several scans."""
    from conftest import SCAN_GROUP_CASES, scan_group_pair
    files, want = [], []
    for case in SCAN_GROUP_CASES:
        multi, inter = scan_group_pair(case)
        files += [multi, inter]
        want += [oracle.decode(multi), oracle.decode(inter)]
    n_pairs = len(files)
    for extra in (J.synth.encode_jpeg(J.synth.synth_image(1280, 720, seed=4), "4:2:0", 88, 0, scans="y|cbcr"),
                  J.synth.encode_jpeg(J.synth.synth_image(640, 480, seed=5), "4:2:2", 80, 40, scans="ycb|cr", optimise=True),
                  scans_per_component_pair(SCANS_PER_COMPONENT_CASES[2])[0]):
        files.append(extra)
        want.append(oracle.decode(extra))
    b = ctx.batch(files, flags)
    b.decode()
    for i, o in enumerate(want):
        assert b.status(i) == 0, i
        assert np.array_equal(b.coefs(i), o.coefs), f"file {i}: coefficients"
        out = b.rgb(i)
        assert np.array_equal(out[..., :3], o.rgb), f"file {i}: RGB"
        if i % 2 == 0 or i >= n_pairs:
            for c, p in enumerate(o.planes):
                assert np.array_equal(b.plane(i, c), p), f"file {i}: plane {c}"
        if i % 2 == 0 and i < n_pairs:
            assert np.array_equal(o.rgb, want[i + 1].rgb)
    b.close()


def test_dropin_api_on_a_file_of_two_scans(ctx, oracle):
    """decode_jpeg_buffer + ycbcr_rgb on a frame whose luma has a scan of its own and whose chroma
    components share one: planes at the Image's strides, RGB the restatement's."""
    from conftest import SCAN_GROUP_CASES, scan_group_pair
    lib = J.load_library()
    data, _ = scan_group_pair(SCAN_GROUP_CASES[1])
    o = oracle.decode(data)
    buf = C.create_string_buffer(data, len(data) + 16)
    fh = _FrameHeader()
    fh.process = -1
    img = C.POINTER(_Image)()
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_FrameHeader), C.POINTER(C.POINTER(_Image))]
    assert lib.decode_jpeg_buffer(buf, len(data), C.byref(fh), C.byref(img)) == 0
    H, W = o.rgb.shape[:2]
    for c in range(3):
        p = o.planes[c]
        stride = img.contents.width[c]
        got = np.ctypeslib.as_array(img.contents.buf[c], shape=(p.shape[0], stride))
        assert np.array_equal(got[:, :p.shape[1]], p), f"Image.buf[{c}] differs"
    pixels = np.zeros((H, 3 * W), np.uint8)
    lib.ycbcr_rgb.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    assert lib.ycbcr_rgb(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
    assert np.array_equal(pixels.reshape(H, W, 3), o.rgb)
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_img(img)
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    lib.free_frame_header(C.byref(fh))


def test_no_restart_long_scan_cross_sequence(ctx, oracle):
    """A scan without restart markers that spans many sequences: exercises the
    cross-sequence fix-up and the second write kernel."""
    data = J.synth.encode_jpeg(J.synth.synth_image(1024, 768, 3), "4:2:0", 85, 0)
    o = oracle.decode(data)
    b = ctx.batch([data], 0)
    b.decode()
    st = b.entropy_stats()
    assert st["multi_seq_intervals"] == 1 and st["sequences"] > 4, st
    _check_image(b, 0, o.coefs, None, o.rgb, "1024x768 no-RST")
    b.close()


def test_small_files_without_restart_markers_in_a_large_batch(ctx, oracle):
    """The lane length follows the size of the whole batch: beside large files, small ones
    coded as one interval are cut into two to four sequences of their own lane length
    (a file of 1 to 70 KB takes every branch of that rule).  Each is checked against the
    restatement, and decodes to the same bytes as alone."""
    big = [J.synth.config_jpeg("C3", seed=s) for s in (2, 3, 4, 5)]          # 66 Mbit: lanes of 1024 bits
    small = []
    for i, (w, h, samp, q) in enumerate([(64, 48, "4:2:0", 60), (160, 120, "4:2:0", 75), (240, 160, "4:4:4", 70),
                                         (320, 240, "4:2:2", 75), (400, 300, "4:2:0", 85), (640, 480, "4:2:0", 50),
                                         (640, 480, "4:2:2", 75), (640, 480, "4:4:4", 85), (800, 600, "4:2:0", 90)]):
        small.append(J.synth.encode_jpeg(J.synth.synth_image(w, h, 50 + i), samp, q, 0, optimise=bool(i & 1)))
    grey = J.synth.encode_jpeg(J.synth.synth_image(512, 384, 77, grey=True), "4:4:4", 80, 0)
    files = big[:2] + small + [grey] + big[2:]
    b = ctx.batch(files, 0)
    b.decode()
    st = b.entropy_stats()
    assert st["multi_seq_intervals"] >= 6, st
    for i, data in enumerate(files):
        if 2 <= i < 2 + len(small) + 1:
            o = oracle.decode(data)
            _check_image(b, i, o.coefs, None, o.rgb, f"small file {i} ({len(data)} bytes)")
            alone = ctx.batch([data], 0)
            alone.decode()
            assert np.array_equal(alone.rgb(0), b.rgb(i)) and np.array_equal(alone.coefs(0), b.coefs(i))
            alone.close()
    o = oracle.decode(big[0])
    _check_image(b, 0, o.coefs, None, o.rgb, "large neighbour")
    b.close()


@pytest.mark.parametrize("config", ["C1", "C2", "C3"])
def test_baseline_configs_vs_oracle(ctx, oracle, config):
    """BASELINE configs at full size, one image each, bit-exact against the restatement."""
    data = J.synth.config_jpeg(config, seed=1)
    o = oracle.decode(data)
    b = ctx.batch([data], 0)
    b.decode()
    _check_image(b, 0, o.coefs, None, o.rgb, config)
    b.close()


def test_config4_8k_no_restart(ctx, oracle):
    """8K 4:2:0 without restart markers (129,600 MCUs, one interval).  The reference
    itself misplaces blocks past 65,535 MCUs (decode.h:15), so pixels are checked
    against the restatement, whose MCU counter is 32-bit."""
    data = J.synth.config_jpeg("C4", seed=0)
    o = oracle.decode(data)
    b = ctx.batch([data], 0)
    b.decode()
    _check_image(b, 0, o.coefs, None, o.rgb, "C4")
    b.close()


def test_batch_is_order_independent_and_matches_single(ctx):
    """Size-independent property at BASELINE shape: an image decodes to the same bytes
    alone, in a batch, and at another position of the batch."""
    files = [J.synth.config_jpeg("C3", seed=s) for s in range(3)]
    files.append(J.synth.config_jpeg("C2", seed=9))
    b = ctx.batch(files, 0)
    b.decode()
    digests = [hashlib.sha256(b.rgb(i).tobytes()).hexdigest() for i in range(len(files))]
    cdig = [hashlib.sha256(b.coefs(i).tobytes()).hexdigest() for i in range(len(files))]
    b.close()
    b2 = ctx.batch(files[::-1], 0)
    b2.decode()
    n = len(files)
    for i in range(n):
        assert hashlib.sha256(b2.rgb(n - 1 - i).tobytes()).hexdigest() == digests[i]
        assert hashlib.sha256(b2.coefs(n - 1 - i).tobytes()).hexdigest() == cdig[i]
    b2.close()
    b3 = ctx.batch([files[1]], 0)
    b3.decode()
    assert hashlib.sha256(b3.rgb(0).tobytes()).hexdigest() == digests[1]
    b3.close()


class _Component(C.Structure):
    _fields_ = [("id", C.c_uint8), ("hsf", C.c_uint8), ("vsf", C.c_uint8), ("x", C.c_uint16), ("y", C.c_uint16),
                ("qtid", C.c_uint8)]


class _FrameHeader(C.Structure):
    _fields_ = [("process", C.c_int), ("precision", C.c_uint8), ("X", C.c_uint16), ("Y", C.c_uint16),
                ("ncs", C.c_uint8), ("cs", C.POINTER(_Component))]


class _Image(C.Structure):
    _fields_ = [("buf", C.POINTER(C.POINTER(C.c_uint8))), ("width", C.POINTER(C.c_int)), ("mcu", C.c_uint16),
                ("n_du", C.c_uint8)]


@pytest.mark.parametrize("name", ["c420_q85_100x75_ri3", "c444_q75_opt_57x43", "g_q95_opt_120x88_rirow",
                                  "C1_c420_q85_512x512"])
def test_dropin_api_like_the_reference_viewer(ctx, name):
    """decode_jpeg_buffer + ycbcr_rgb/y_rgb + free_img, called the way the reference's
    main() calls them (jpeg_decoder.c:52-104), with a pitch wider than 3*width."""
    lib = J.load_library()
    data, g = load_fixture(name)
    buf = C.create_string_buffer(data, len(data) + 16)
    fh = _FrameHeader()
    fh.process = -1
    img = C.POINTER(_Image)()
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_FrameHeader), C.POINTER(C.POINTER(_Image))]
    assert lib.decode_jpeg_buffer(buf, len(data), C.byref(fh), C.byref(img)) == 0
    H, W = g["rgb"].shape[:2]
    assert (fh.X, fh.Y) == (W, H)
    for c in range(fh.ncs):
        p = g[f"plane{c}"]
        got = np.ctypeslib.as_array(img.contents.buf[c], shape=p.shape)
        assert img.contents.width[c] == p.shape[1]
        assert np.array_equal(got, p), f"Image.buf[{c}] differs"
    pitch = 3 * W + 8
    pixels = np.zeros((H, pitch), np.uint8)
    fn = lib.y_rgb if fh.ncs == 1 else lib.ycbcr_rgb
    fn.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    assert fn(C.byref(fh), img, pixels.ctypes.data, W, H, pitch) == 0
    assert np.array_equal(pixels[:, :3 * W].reshape(H, W, 3), g["rgb"])
    assert not pixels[:, 3 * W:].any(), "bytes beyond the row were touched"
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_img(img)
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    lib.free_frame_header(C.byref(fh))


def test_dropin_api_on_a_file_of_one_scan_per_component(ctx, oracle):
    """decode_jpeg_buffer + ycbcr_rgb on a frame coded in three scans: the Image's planes (at
    the Image's own stride, padded to whole MCUs) hold the visible samples of every component
    and the RGB is the restatement's.  The reference itself has no output for such a file."""
    lib = J.load_library()
    data, _ = scans_per_component_pair(SCANS_PER_COMPONENT_CASES[1])        # 100 x 75 4:2:0, fitted tables
    o = oracle.decode(data)
    buf = C.create_string_buffer(data, len(data) + 16)
    fh = _FrameHeader()
    fh.process = -1
    img = C.POINTER(_Image)()
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_FrameHeader), C.POINTER(C.POINTER(_Image))]
    assert lib.decode_jpeg_buffer(buf, len(data), C.byref(fh), C.byref(img)) == 0
    H, W = o.rgb.shape[:2]
    assert (fh.X, fh.Y, fh.ncs) == (W, H, 3)
    for c in range(3):
        p = o.planes[c]
        stride = img.contents.width[c]
        assert stride >= p.shape[1]
        got = np.ctypeslib.as_array(img.contents.buf[c], shape=(p.shape[0], stride))
        assert np.array_equal(got[:, :p.shape[1]], p), f"Image.buf[{c}] differs"
    pixels = np.zeros((H, 3 * W), np.uint8)
    lib.ycbcr_rgb.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    assert lib.ycbcr_rgb(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
    assert np.array_equal(pixels.reshape(H, W, 3), o.rgb)
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_img(img)
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    lib.free_frame_header(C.byref(fh))


def _frame_of(g, hsf, vsf):
    """_FrameHeader + component array for a golden fixture's geometry."""
    H, W = g["rgb"].shape[:2]
    ncs = len(hsf)
    cs = (_Component * ncs)()
    hm, vm = max(hsf), max(vsf)
    for c in range(ncs):
        cs[c].id, cs[c].hsf, cs[c].vsf, cs[c].qtid = c + 1, hsf[c], vsf[c], 0
        cs[c].x = -(-W * hsf[c] // hm)
        cs[c].y = -(-H * vsf[c] // vm)
    fh = _FrameHeader()
    fh.process, fh.precision, fh.X, fh.Y, fh.ncs = 0, 8, W, H, ncs
    fh.cs = C.cast(cs, C.POINTER(_Component))
    return fh, cs


@pytest.mark.parametrize("name,hsf,vsf", [
    ("c420_q85_100x75_ri3", (2, 1, 1), (2, 1, 1)), ("c422_q75_120x72", (2, 1, 1), (1, 1, 1)),
    ("c444_q75_opt_57x43", (1, 1, 1), (1, 1, 1)), ("g_q95_opt_120x88_rirow", (1,), (1,)),
    ("C1_c420_q85_512x512", (2, 1, 1), (2, 1, 1)), ("c422_q50_33x17", (2, 1, 1), (1, 1, 1))])
def test_colour_entry_points_convert_the_planes_they_are_given(ctx, name, hsf, vsf):
    """ycbcr_rgb / y_rgb on an Image the caller filled (allocate_img + the reference's own
    planes from the golden file): the RGB is the reference's (colour_conversion.c:5-83), at a
    pitch wider than a row; an Image whose planes the caller edits after a decode gets the
    edited samples converted, not the RGB of the decode."""
    lib = J.load_library()
    _, g = load_fixture(name)
    fh, cs = _frame_of(g, hsf, vsf)
    H, W = g["rgb"].shape[:2]
    lib.allocate_img.argtypes = [C.POINTER(_FrameHeader)]
    lib.allocate_img.restype = C.POINTER(_Image)
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    fn = lib.y_rgb if len(hsf) == 1 else lib.ycbcr_rgb
    fn.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    img = lib.allocate_img(C.byref(fh))
    assert img
    for c in range(len(hsf)):
        p = np.ascontiguousarray(g[f"plane{c}"])
        assert img.contents.width[c] == p.shape[1]
        C.memmove(img.contents.buf[c], p.ctypes.data, p.size)
    pitch = 3 * W + 8
    pixels = np.zeros((H, pitch), np.uint8)
    assert fn(C.byref(fh), img, pixels.ctypes.data, W, H, pitch) == 0
    assert np.array_equal(pixels[:, :3 * W].reshape(H, W, 3), g["rgb"])
    assert not pixels[:, 3 * W:].any(), "bytes beyond the row were touched"
    # a window of the frame: the top-left corner, as the reference's loops would produce it
    w2, h2 = max(1, W // 2 - 1), max(1, H // 3)
    win = np.zeros((h2, 3 * w2), np.uint8)
    assert fn(C.byref(fh), img, win.ctypes.data, w2, h2, 3 * w2) == 0
    assert np.array_equal(win.reshape(h2, w2, 3), g["rgb"][:h2, :w2])
    lib.free_img(img)


@pytest.mark.parametrize("sf", [((1, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((3, 1), (1, 1), (1, 1)), ((4, 2), (2, 1), (1, 2)),
                                ((3, 3), (1, 3), (3, 1)), ((2, 4), (1, 1), (2, 2)), ((4, 4), (1, 1), (1, 1))])
@pytest.mark.parametrize("size", [(37, 23), (256, 64), (1001, 5)])
def test_planes_to_rgb_at_any_sampling_factors(ctx, oracle, sf, size):
    """mj_planes_to_rgb against the reference's indexing and arithmetic (colour_conversion.c:44-79:
    source sample (uint32)(ratio * i) with ratio = (float) sf / max_sf; the colour step through the
    restatement) for sampling factors the decode path itself never produces, 3 included, where the
    float product decides which sample a pixel takes."""
    lib = J.load_library()
    W, H = size
    hs, vs = [f[0] for f in sf], [f[1] for f in sf]
    hm, vm = max(hs), max(vs)
    rng = np.random.default_rng(W * 7 + H + hs[0] * 100 + vs[0])
    planes, strides = [], []
    for c in range(3):
        x, y = -(-W * hs[c] // hm), -(-H * vs[c] // vm)
        pw = -(-x // (8 * hs[c])) * 8 * hs[c]
        ph = -(-y // (8 * vs[c])) * 8 * vs[c]
        planes.append(rng.integers(0, 256, (ph, pw), dtype=np.uint8))
        strides.append(pw)
    # the reference's loops in numpy: float32 ratios, float32 products, truncation
    i = np.arange(H, dtype=np.float32)
    j = np.arange(W, dtype=np.float32)
    comp = []
    for c in range(3):
        hr, vr = np.float32(hs[c]) / np.float32(hm), np.float32(vs[c]) / np.float32(vm)
        rows = (i * vr).astype(np.uint32)
        cols = (hr * j).astype(np.uint32)
        comp.append(planes[c][rows][:, cols])
    want = oracle.ycbcr_row(comp[0].ravel(), comp[1].ravel(), comp[2].ravel()).reshape(H, W, 3)
    pitch = 3 * W + 5
    out = np.zeros((H, pitch), np.uint8)
    lib.mj_planes_to_rgb.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    pp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    rc = lib.mj_planes_to_rgb(ctx.h, 3, pp, (C.c_int32 * 3)(*strides), (C.c_int32 * 3)(*hs), (C.c_int32 * 3)(*vs), W, H,
                              out.ctypes.data, pitch)
    assert rc == 0, lib.mj_last_error()
    assert np.array_equal(out[:, :3 * W].reshape(H, W, 3), want)
    assert not out[:, 3 * W:].any()
    # one component: R = G = B at the stride given
    grey = np.zeros((H, 3 * W), np.uint8)
    rc = lib.mj_planes_to_rgb(ctx.h, 1, pp, (C.c_int32 * 3)(*strides), (C.c_int32 * 3)(1, 1, 1), (C.c_int32 * 3)(1, 1, 1), min(W, strides[0]),
                              min(H, planes[0].shape[0]), grey.ctypes.data, 3 * W)
    assert rc == 0
    w1, h1 = min(W, strides[0]), min(H, planes[0].shape[0])
    assert np.array_equal(grey.reshape(H, W, 3)[:h1, :w1], np.repeat(planes[0][:h1, :w1, None], 3, 2))


def test_colour_entry_point_sees_planes_edited_after_the_decode(ctx, oracle):
    lib = J.load_library()
    data, g = load_fixture("c420_q85_100x75_ri3")
    buf = C.create_string_buffer(data, len(data) + 16)
    fh = _FrameHeader()
    fh.process = -1
    img = C.POINTER(_Image)()
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_FrameHeader), C.POINTER(C.POINTER(_Image))]
    assert lib.decode_jpeg_buffer(buf, len(data), C.byref(fh), C.byref(img)) == 0
    H, W = g["rgb"].shape[:2]
    lib.ycbcr_rgb.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    pixels = np.zeros((H, 3 * W), np.uint8)
    assert lib.ycbcr_rgb(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
    assert np.array_equal(pixels.reshape(H, W, 3), g["rgb"])                   # untouched: the decode's own RGB
    # the caller inverts the luma plane and flattens Cb: the next call converts those samples
    planes = [np.ctypeslib.as_array(img.contents.buf[c], shape=g[f"plane{c}"].shape) for c in range(3)]
    planes[0][:] = 255 - planes[0]
    planes[1][:] = 77
    cbu = np.repeat(np.repeat(planes[1], 2, 0), 2, 1)[:H, :W]
    cru = np.repeat(np.repeat(planes[2], 2, 0), 2, 1)[:H, :W]
    want = oracle.ycbcr_row(planes[0][:H, :W].ravel(), cbu.ravel(), cru.ravel()).reshape(H, W, 3)
    assert lib.ycbcr_rgb(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
    assert np.array_equal(pixels.reshape(H, W, 3), want)
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_img(img)
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    lib.free_frame_header(C.byref(fh))


class _Bitstream(C.Structure):
    _fields_ = [("encoded_data", C.c_void_p), ("offset", C.c_uint8), ("size", C.c_uint32)]


def _parse_up_to_sos(lib, buf, data):
    """The marker walk of decode.c:160-341 up to the scan header, with the library's own segment
    parsers: (fh, sh, hts, qts, ri, cursor on the first entropy-coded byte)."""
    base = C.addressof(buf)
    fh = _FrameHeader()
    fh.process = -1
    sh, hts, qts = (C.create_string_buffer(4096) for _ in range(3))     # ScanHeader, HuffTables, QuantTables: opaque here
    ri = C.c_uint16(0)
    bs = _Bitstream()
    have_hts = False
    i = 2
    while True:
        assert data[i] == 0xFF
        m, n = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        bs.encoded_data, bs.offset = base + i + 2, 0
        if m == 0xC0:
            assert lib.decode_frame_header(0, C.byref(bs), C.byref(fh)) == 0
        elif m == 0xC4:
            if not have_hts:
                lib.new_huff_tables(fh.process, hts)
                have_hts = True
            assert lib.decode_huff_tables(C.byref(bs), hts) == 0
        elif m == 0xDB:
            assert lib.decode_quant_table(C.byref(bs), qts) == 0
        elif m == 0xDD:
            assert lib.decode_restart_interval(C.byref(bs), C.byref(ri)) == 0
        elif m == 0xDA:
            assert lib.decode_scan_header(C.byref(bs), sh) == 0
            bs.encoded_data, bs.offset = base + i + 2 + n, 0
            return fh, sh, hts, qts, ri, bs
        i += 2 + n


@pytest.mark.parametrize("name", ["c420_q85_100x75_ri3", "g_q75_72x40_ri4", "c444_q75_opt_57x43"])
def test_decode_scan_called_on_its_own(ctx, name):
    """The inner seam (decode.c:333: decode_scan at the SOS case) with structs from the segment
    parsers, outside decode_jpeg_buffer: planes as the reference's, the cursor left on the marker
    that ends the scan, and the search for that marker bounded by mj_set_scan_limit."""
    lib = J.load_library()
    data, g = load_fixture(name)
    buf = C.create_string_buffer(data, len(data) + 16)
    for f in (lib.decode_frame_header, lib.decode_huff_tables, lib.decode_quant_table, lib.decode_restart_interval,
              lib.decode_scan_header, lib.new_huff_tables, lib.decode_scan, lib.free_huff_tables):
        f.restype = C.c_int
    lib.decode_frame_header.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.new_huff_tables.argtypes = [C.c_int, C.c_void_p]
    for f in (lib.decode_huff_tables, lib.decode_quant_table, lib.decode_restart_interval, lib.decode_scan_header):
        f.argtypes = [C.c_void_p, C.c_void_p]
    fh, sh, hts, qts, ri, bs = _parse_up_to_sos(lib, buf, data)
    lib.allocate_img.argtypes = [C.POINTER(_FrameHeader)]
    lib.allocate_img.restype = C.POINTER(_Image)
    img = lib.allocate_img(C.byref(fh))
    assert img
    lib.decode_scan.argtypes = [C.c_void_p, C.POINTER(_Image), C.POINTER(_FrameHeader), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16]
    lib.mj_set_scan_limit.argtypes = [C.c_size_t]
    lib.mj_set_scan_limit.restype = None
    start = bs.encoded_data
    lib.mj_set_scan_limit(16)                      # the scan is longer than that: refused, nothing read beyond
    assert lib.decode_scan(C.byref(bs), img, C.byref(fh), sh, hts, qts, ri.value) == -1
    lib.mj_set_scan_limit(0)                       # back to the default
    bs.encoded_data, bs.offset = start, 0
    assert lib.decode_scan(C.byref(bs), img, C.byref(fh), sh, hts, qts, ri.value) == 0
    end = bs.encoded_data - C.addressof(buf)
    assert data[end:end + 2] == b"\xff\xd9" and bs.offset == 0
    for c in range(fh.ncs):
        p = g[f"plane{c}"]
        assert np.array_equal(np.ctypeslib.as_array(img.contents.buf[c], shape=p.shape), p), f"Image.buf[{c}] differs"
    H, W = g["rgb"].shape[:2]
    fn = lib.y_rgb if fh.ncs == 1 else lib.ycbcr_rgb
    fn.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    pixels = np.zeros((H, 3 * W), np.uint8)
    assert fn(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
    assert np.array_equal(pixels.reshape(H, W, 3), g["rgb"])
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_img(img)
    lib.free_huff_tables.argtypes = [C.c_void_p]
    lib.free_huff_tables(hts)
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    lib.free_frame_header(C.byref(fh))


def _entropy_start(data):
    i = data.index(b"\xff\xda")
    return i + 2 + ((data[i + 2] << 8) | data[i + 3])


def _damaged_variants(data, seed):
    """Truncations, bit flips, stray FF bytes and an extra RSTn inside the entropy-coded
    segment: what a decoder meets on a cut-off transfer or a bad sector."""
    rng = np.random.default_rng(seed)
    h = _entropy_start(data)
    n = len(data) - h
    out = []
    for frac in (0.0, 0.3, 0.8):
        out.append(("truncated", data[:h + int(n * frac)]))
    for _ in range(4):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            d[h + int(rng.integers(0, n - 2))] ^= 1 << int(rng.integers(0, 8))
        out.append(("bitflips", bytes(d)))
    d = bytearray(data)
    for _ in range(3):
        d[h + int(rng.integers(0, n - 2))] = 0xFF
    out.append(("stray_ff", bytes(d)))
    d = bytearray(data)
    p = h + int(rng.integers(0, n - 4))
    d[p], d[p + 1] = 0xFF, 0xD0 + int(rng.integers(0, 8))
    out.append(("extra_rst", bytes(d)))
    return out


@pytest.mark.parametrize("flags", [0, binding.MJ_ENTROPY_SERIAL])
def test_damaged_streams_neither_fault_nor_disturb_their_neighbours(ctx, flags):
    """The reference reads past the end of a damaged file (decode.c:578-589 has no bound);
    there is nothing to be equal to.  The contract here: every access stays inside the
    batch's buffers, a stream that ends early is reported in the image's status word, and
    the intact images of the same batch are decoded exactly as if they were alone."""
    names = ["c420_q85_100x75_ri3", "c444_q85_80x56_rirow", "g_q75_72x40_ri4", "c422_q95_opt_136x40_ri5"]
    files, expect = [], []
    for k, name in enumerate(names):
        d, g = load_fixture(name)
        files.append(d)
        expect.append(g)
        for kind, bad in _damaged_variants(d, 100 + k):
            files.append(bad)
            expect.append(kind)
    d, g = load_fixture("C1_c420_q85_512x512")
    for kind, bad in _damaged_variants(d, 7)[:5]:
        files.append(bad)
        expect.append(kind)
    files.append(d)
    expect.append(g)
    b = ctx.batch(files, flags)
    for _ in range(2):
        b.decode()
        b.sync()
        for i, e in enumerate(expect):
            if isinstance(e, str):
                st = b.status(i)
                if e == "truncated":
                    assert st != 0, f"image {i}: a stream cut short must be reported"
                b.rgb(i)    # readable, whatever it holds
            else:
                _check_image(b, i, e["coefs"], None, e["rgb"], f"intact image {i}")
    b.close()


TINY_AND_THIN = [
    # (w, h, sampling, quality, restart interval, grey): the smallest images, single rows and
    # columns of MCUs, and the widest and tallest an MCU row / column can get
    (1, 1, "4:2:0", 85, 0, False),
    (8, 1, None, 85, 0, True),
    (7, 9, "4:4:4", 90, 0, False),
    (8, 8, "4:2:2", 75, 1, False),
    (17, 1, "4:2:0", 85, 0, False),
    (1, 17, "4:4:0", 85, 1, False),
    (16, 16, "4:1:1", 60, 0, False),
    (65535, 8, "4:2:0", 70, 512, False),
    (9, 65535 // 8, "4:2:2", 70, 0, False),
    (65528, 3, None, 85, 0, True),
]


@pytest.mark.parametrize("case", TINY_AND_THIN, ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}_ri{c[4]}")
def test_tiny_and_extreme_shapes_vs_oracle(ctx, oracle, case):
    w, h, samp, q, ri, grey = case
    assert not (grey and w % 8), "grey widths that are not multiples of 8 hit the reference's stride quirk (colour_conversion.c:20)"
    img = J.synth.synth_image(w, h, seed=w + 3 * h, grey=grey)
    data = J.synth.encode_jpeg(img, samp or "4:4:4", q, ri)
    o = oracle.decode(data)
    b = ctx.batch([data], binding.MJ_WANT_PLANES)
    b.decode()
    _check_image(b, 0, o.coefs, o.planes, o.rgb, str(case))
    b.close()


def test_empty_and_unparsable_inputs_are_refused(ctx):
    """No images, an empty buffer, bytes that are not a JPEG, and a progressive file: the
    batch is refused with an error (decode_jpeg_buffer returns -1 for these, decode.c:138-424),
    and the context stays usable."""
    good, g = load_fixture("c420_q85_64x48")
    for files in ([], [b""], [b"\x00" * 100], [good, b"\xff\xd8\xff\xd9"], [good.replace(b"\xff\xc0", b"\xff\xc2", 1)]):
        with pytest.raises(RuntimeError):
            ctx.batch(files, 0)
    b = ctx.batch([good], 0)
    b.decode()
    _check_image(b, 0, g["coefs"], None, g["rgb"], "after refusals")
    b.close()


def test_rgba_output_is_rgb_plus_opaque_alpha(ctx):
    """MJ_OUT_RGBA: the same pixels, four bytes each with alpha 255, rows of 4 * X bytes
    (grey, every subsampling, widths that are not multiples of four)."""
    names = ["c420_q85_100x75_ri3", "c422_q50_33x17", "c444_q75_opt_57x43", "g_q85_edges_48x48", "c420_q100_edges_40x40",
             "C1_c420_q85_512x512"]
    files, golds = zip(*[load_fixture(n) for n in names])
    b = ctx.batch(list(files), binding.MJ_OUT_RGBA)
    b.decode()
    for i, (name, g) in enumerate(zip(names, golds)):
        out = b.rgb(i)
        assert out.shape == g["rgb"].shape[:2] + (4,), name
        assert b.info(i).rgb_pitch == 4 * g["rgb"].shape[1], name
        assert np.array_equal(out[..., :3], g["rgb"]), name
        assert (out[..., 3] == 255).all(), name
    b.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_images_vs_oracle(ctx, oracle, seed):
    """Seeded random shapes, samplings, qualities, restart intervals and noise levels in
    mixed batches, both entropy modes: coefficients, planes and RGB bit-exact against the
    CPU restatement.  (The same loop has been run over many thousands of images.)"""
    rng = np.random.default_rng(seed)
    samplings = ["4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1"]
    files, meta = [], []
    for _ in range(24):
        grey = rng.random() < 0.15
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        if rng.random() < 0.08:
            w, h = int(rng.integers(700, 2200)), int(rng.integers(500, 1500))
        if grey:
            w = max(8, w // 8 * 8)
        samp = samplings[int(rng.integers(0, len(samplings)))]
        q = int(rng.integers(15, 101))
        ri = int(rng.choice([0, 0, 1, 2, 3, 7, 16, 40, 200]))
        sigma = float(rng.choice([0.0, 2.0, 12.0, 40.0]))
        img = J.synth.synth_image(w, h, seed=int(rng.integers(0, 1 << 30)), sigma=sigma, grey=grey)
        files.append(J.synth.encode_jpeg(img, samp, q, ri))
        meta.append((w, h, samp, q, ri, grey, sigma))
    wants = [oracle.decode(d) for d in files]
    for flags in (0, binding.MJ_ENTROPY_SERIAL):
        b = ctx.batch(files, binding.MJ_WANT_PLANES | flags)
        b.decode()
        for i, o in enumerate(wants):
            _check_image(b, i, o.coefs, o.planes, o.rgb, f"{meta[i]} flags={flags}")
        b.close()


def _fnv1a(data):
    h = 1469598103934665603
    for byte in data:
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_headless_jpeg_decoder_executable_on_config1():
    """BASELINE config 1: the 512x512 4:2:0 file decoded by the jpeg_decoder executable (the
    reference viewer's call sequence, jpeg_decoder.c:44-104, without the SDL window) in a
    child process; the checksum it prints is that of the reference's RGB."""
    import os
    import re
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "jpeg-decoder_amd", "jpeg_decoder")
    assert os.path.exists(exe), "jpeg_decoder is not built (run __graft_entry__.build())"
    _, g = load_fixture("C1_c420_q85_512x512")
    r = subprocess.run([exe, os.path.join(GOLDEN, "C1_c420_q85_512x512.jpg"), "-n", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+)x(\d+) rgb24 fnv1a=([0-9a-f]{16})", r.stdout)
    assert m, r.stdout
    assert (int(m.group(1)), int(m.group(2))) == (512, 512)
    assert int(m.group(3), 16) == _fnv1a(np.ascontiguousarray(g["rgb"]).tobytes())


def test_many_live_images_through_the_dropin_entry_points(ctx):
    """More Images alive at once than any fixed table would hold: each keeps its own GPU
    result until free_img (the reference has no such limit either, decode.c:95-136)."""
    lib = J.load_library()
    data, g = load_fixture("c420_q85_64x48")
    buf = C.create_string_buffer(data, len(data) + 16)
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_FrameHeader), C.POINTER(C.POINTER(_Image))]
    lib.ycbcr_rgb.argtypes = [C.POINTER(_FrameHeader), C.POINTER(_Image), C.c_void_p, C.c_uint16, C.c_uint16, C.c_uint16]
    lib.free_img.argtypes = [C.POINTER(_Image)]
    lib.free_img.restype = None
    lib.free_frame_header.argtypes = [C.POINTER(_FrameHeader)]
    live = []
    for _ in range(80):
        fh = _FrameHeader()
        fh.process = -1
        img = C.POINTER(_Image)()
        assert lib.decode_jpeg_buffer(buf, len(data), C.byref(fh), C.byref(img)) == 0
        live.append((fh, img))
    H, W = g["rgb"].shape[:2]
    for fh, img in live[::-1]:
        pixels = np.zeros((H, 3 * W), np.uint8)
        assert lib.ycbcr_rgb(C.byref(fh), img, pixels.ctypes.data, W, H, 3 * W) == 0
        assert np.array_equal(pixels.reshape(H, W, 3), g["rgb"])
        lib.free_img(img)
        lib.free_frame_header(C.byref(fh))


@pytest.mark.parametrize("config", ["C2", "C3"])
def test_full_batch_of_64_every_image_vs_oracle(ctx, config):
    """The benchmark's batch at full size -- 64 images of BASELINE config 2 / 3 in one
    batch -- with every image's coefficient blocks and RGB compared (SHA-256) against the
    CPU restatement, which tests/test_oracle_vs_ref.py pins to the compiled reference at
    these sizes."""
    import multiprocessing as mp
    from _workers import config_file_and_digests, pool_size
    with mp.get_context("spawn").Pool(pool_size()) as pool:
        items = pool.map(config_file_and_digests, [(config, 1000 + s) for s in range(64)], chunksize=1)
    b = ctx.batch([it[0] for it in items], 0)
    b.decode()
    b.sync()
    bad = []
    for i, (_, dc, dr) in enumerate(items):
        if b.status(i) != 0 or hashlib.sha256(b.coefs(i).tobytes()).hexdigest() != dc or \
                hashlib.sha256(b.rgb(i).tobytes()).hexdigest() != dr:
            bad.append(i)
    b.close()
    assert not bad, f"{config}: images {bad} differ from the restatement"


def test_batch_large_enough_for_the_longest_lanes(ctx, oracle):
    """The synchronisation kernel takes its third shape (a sequence per quarter wavefront, two
    groups of sequences per workgroup) when the workgroup groups that are sure to hold work
    number eight per compute unit or more.  Seventeen 512 x 512 files with a restart interval
    per MCU (1,024 intervals = 128 groups each) make such a batch without much data; beside
    them files whose intervals are long enough for rounds (4K, an interval per MCU row), one
    scan without restart markers (cross-sequence pass), a grey and a 4:4:4 file.  Every image
    is compared with the restatement."""
    import torch
    tiny = [J.synth.encode_jpeg(J.synth.synth_image(512, 512, 900 + s), "4:2:0", 80, 1) for s in range(17)]
    assert 17 * (1024 // 8) >= 8 * torch.cuda.get_device_properties(0).multi_processor_count
    rows = [J.synth.config_jpeg("C3", seed=s) for s in (31, 32)]
    long_scan = J.synth.encode_jpeg(J.synth.synth_image(1024, 768, 5), "4:2:0", 85, 0)
    grey = J.synth.encode_jpeg(J.synth.synth_image(640, 480, 6, grey=True), "4:4:4", 75, J.synth.mcus_per_row(640, "4:4:4", True))
    full = J.synth.encode_jpeg(J.synth.synth_image(1920, 1080, 7), "4:4:4", 85, J.synth.mcus_per_row(1920, "4:4:4", False), optimise=True)
    files = tiny[:9] + [rows[0], long_scan] + tiny[9:] + [grey, rows[1], full]
    b = ctx.batch(files, 0)
    b.decode()
    b.sync()
    st = b.entropy_stats()
    assert st["multi_seq_intervals"] >= 1, st
    for i, data in enumerate(files):
        o = oracle.decode(data)
        assert b.status(i) == 0
        _check_image(b, i, o.coefs, None, o.rgb, f"file {i} ({len(data)} bytes)")
    b.close()


def test_many_files_in_one_call_are_prepared_by_several_threads(ctx, oracle):
    """A call of 64 files or more parses its headers and flattens its tables on several host
    threads (each with the table sets it meets, merged by key; files coded in one scan per
    component last, by the caller's thread).  140 small files -- Annex K tables, tables fitted
    to the image, grey, two of them coded in one scan per component, in no order -- decode as
    the restatement says; with two damaged files among them the call is refused and the
    message names the first."""
    files = []
    for i in range(140):
        w, h = 48 + 8 * (i % 7), 40 + 8 * (i % 5)
        grey = (i % 11) == 3
        samp = "4:4:4" if grey else ("4:2:0", "4:2:2", "4:4:4")[i % 3]
        img = J.synth.synth_image(w, h, 3000 + i, grey=grey)
        if i in (37, 101):
            files.append(J.synth.encode_jpeg(img, "4:2:0", 80, 0, scans_per_component=True))
        else:
            files.append(J.synth.encode_jpeg(img, samp, 50 + (i % 5) * 10, (i % 4) * 3, optimise=bool(i % 2)))
    b = ctx.batch(files, 0)
    b.decode()
    b.sync()
    for i, data in enumerate(files):
        o = oracle.decode(data)
        assert b.status(i) == 0
        assert np.array_equal(b.rgb(i), o.rgb), f"file {i}: RGB"
        if i not in (37, 101):
            assert np.array_equal(b.coefs(i), o.coefs), f"file {i}: coefficient blocks"
    b.close()
    bad = list(files)
    bad[77] = bad[77][:2] + b"\xff\xc2" + bad[77][4:]          # not a baseline frame header where one is expected
    bad[120] = b"\x00" * 64
    with pytest.raises(RuntimeError, match="image 77"):
        ctx.batch(bad, 0)
    b = ctx.batch(files[:70], 0)                                 # the context is still usable
    b.decode()
    assert np.array_equal(b.rgb(5), oracle.decode(files[5]).rgb)
    b.close()


C5_CASES = [(mode, q, opt) for mode in ("grey", "4:2:2", "4:2:0") for q in (50, 75, 95) for opt in (False, True)]


def test_config5_mixed_batch_1080p_vs_oracle(ctx, oracle):
    """BASELINE config 5 at 1920x1080: grey / 4:2:2 / 4:2:0 x q 50 / 75 / 95 x Annex K tables /
    tables fitted to the image (T.81 Annex K.2), half of them with one restart
    interval per MCU row, all in one batch; coefficients, planes and RGB bit-exact against the
    CPU restatement.  (The compiled reference cannot check this mix: at q95 with restart markers
    it runs into the marker bytes and crashes, bitstream.c:84-176; the q50 / q75 files without
    markers are pinned to it in tests/test_oracle_vs_ref.py.)"""
    files, wants = [], []
    for k, (mode, q, opt) in enumerate(C5_CASES):
        grey = mode == "grey"
        img = J.synth.synth_image(1920, 1080, seed=500 + k, grey=grey)
        samp = "4:4:4" if grey else mode
        ri = J.synth.mcus_per_row(1920, samp, grey) if k % 2 else 0
        files.append(J.synth.encode_jpeg(img, samp, q, ri, optimise=opt))
        wants.append(oracle.decode(files[-1]))
    b = ctx.batch(files, binding.MJ_WANT_PLANES)
    b.decode()
    for i, o in enumerate(wants):
        _check_image(b, i, o.coefs, o.planes, o.rgb, f"C5 {C5_CASES[i]}")
    b.close()
