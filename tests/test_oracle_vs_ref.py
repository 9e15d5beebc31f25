"""Pin the restatement against the compiled reference directly (leaf sweeps).

Needs oracle/_ref/libjpegref.so, which is built from /root/reference/src by
oracle/Makefile in the build container and travels with the tree; skipped where
it is absent.
"""
import numpy as np
import pytest

from conftest import ALL_FIXTURES, load_fixture


def test_colour_exhaustive_all_triples(oracle, ref):
    """All 2^24 (Y,Cb,Cr) triples through colour_conversion.c:66-79."""
    cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(64, dtype=np.uint8), indexing="ij")
    cb = cb.ravel()
    for y in range(256):
        yv = np.full(cb.size, y, np.uint8)
        for q in range(4):
            crq = (cr.ravel() + 64 * q).astype(np.uint8)
            a = ref.ycbcr_row(yv, cb, crq)
            b = oracle.ycbcr_row(yv, cb, crq)
            assert np.array_equal(a, b), f"colour differs at Y={y}"


def test_kernel_colour_arithmetic_all_triples(ref):
    """The integer form of the colour transform that the IDCT + colour kernel uses
    (csrc/mj_colour_core.h: per-chroma offsets for R and B, a 2^-20 fixed-point sum for G, the
    one wrapping triple patched) equals colour_conversion.c:66-79 as compiled, on all 2^24
    (Y, Cb, Cr) triples."""
    import jpeg_decoder_amd as J
    emu = J.Emu()
    cb, cr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(64, dtype=np.uint8), indexing="ij")
    cb = cb.ravel()
    for y in range(256):
        yv = np.full(cb.size, y, np.uint8)
        for q in range(4):      # the reference's row width is 16 bits: a quarter of the (Cb, Cr) plane per call
            crq = (cr.ravel() + 64 * q).astype(np.uint8)
            assert np.array_equal(ref.ycbcr_row(yv, cb, crq), emu.ycbcr_row(yv, cb, crq)), f"colour differs at Y={y}"
    # the kernel recognises the one chroma pair whose G can wrap by the sum of its G shares
    assert emu.lib.mj_emu_wrap_sum_pairs() == 1
    assert tuple(emu.ycbcr_row([219], [228], [28])[0]) == tuple(ref.ycbcr_row([219], [228], [28])[0]) and \
        emu.ycbcr_row([219], [228], [28])[0][1] == 0


def _random_blocks(rng, n):
    blocks = np.zeros((n, 64), np.int16)
    for i in range(n):
        kind = i % 4
        if kind == 0:      # sparse, realistic magnitudes
            nz = rng.integers(1, 12)
            idx = rng.choice(64, nz, replace=False)
            blocks[i, idx] = rng.integers(-60, 60, nz)
            blocks[i, 0] = rng.integers(-200, 200)
        elif kind == 1:    # dense, saturating
            blocks[i] = rng.integers(-1024, 1024, 64)
        elif kind == 2:    # DC + one AC
            blocks[i, 0] = rng.integers(-1024, 1024)
            blocks[i, rng.integers(1, 64)] = rng.integers(-1024, 1024)
        else:              # extremes incl. int16 wrap in the dequant product
            blocks[i] = rng.integers(-2048, 2048, 64)
    return blocks


def test_dequant_idct_random_blocks(oracle, ref):
    rng = np.random.default_rng(1234)
    blocks = _random_blocks(rng, 6000)
    tables = [np.ones(64, np.uint16), rng.integers(1, 40, 64).astype(np.uint16),
              rng.integers(1, 256, 64).astype(np.uint16), np.full(64, 255, np.uint16)]
    for i, zz in enumerate(blocks):
        qt = tables[i % len(tables)]
        assert np.array_equal(ref.dequant_idct(zz, qt), oracle.dequant_idct(zz, qt)), f"block {i}"


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_ref_reproduces_golden(ref, name):
    """The committed goldens are what this build of the reference produces."""
    data, gold = load_fixture(name)
    d = ref.decode(data)
    assert np.array_equal(d.coefs, gold["coefs"])
    assert np.array_equal(d.rgb, gold["rgb"])


@pytest.mark.parametrize("config", ["C2", "C3"])
def test_oracle_equals_reference_at_full_size(oracle, ref, config):
    """BASELINE configs 2 and 3 at full size (1920x1080 4:4:4 and 3840x2160 4:2:0, one
    restart interval per MCU row): coefficients, planes and RGB of the restatement equal
    the compiled reference's."""
    import jpeg_decoder_amd as J
    for seed in (1, 1000):
        data = J.synth.config_jpeg(config, seed)
        a, b = ref.decode(data), oracle.decode(data)
        assert np.array_equal(a.coefs, b.coefs), (config, seed)
        assert all(np.array_equal(x, y) for x, y in zip(a.planes, b.planes)), (config, seed)
        assert np.array_equal(a.rgb, b.rgb), (config, seed)


def test_oracle_equals_reference_coefficients_on_config4(oracle, ref):
    """BASELINE config 4 (7680x4320 4:2:0, no restart markers, 129,600 MCUs): coefficient
    blocks only -- the reference places blocks with a 16-bit MCU counter (decode.h:15,
    decode.c:473): the second half of the picture is written over the first, so it has
    no pixels to compare, while its entropy decode is not affected."""
    import jpeg_decoder_amd as J
    data = J.synth.config_jpeg("C4", 0)
    a, b = ref.decode(data), oracle.decode(data)
    assert a.coefs.shape == b.coefs.shape == (129600 * 6, 64)
    assert np.array_equal(a.coefs, b.coefs)


@pytest.mark.parametrize("mode,q,opt", [("grey", 50, True), ("4:2:2", 75, True), ("4:2:0", 50, True), ("4:2:0", 75, False),
                                        ("4:2:2", 95, True), ("grey", 95, True)])
def test_oracle_equals_reference_on_fitted_tables(oracle, ref, mode, q, opt):
    """Files of the mixed workload (BASELINE config 5) whose Huffman tables are fitted to
    the image (T.81 Annex K.2), without restart markers: the restatement equals the
    compiled reference (the q95 files carry 16-bit codes).  With restart markers at q95
    the reference itself fails whatever the tables are: its restart check looks at a
    partly consumed FF byte (bitstream.c:84-176) and it then decodes marker bytes as data."""
    import jpeg_decoder_amd as J
    grey = mode == "grey"
    img = J.synth.synth_image(640, 480, seed=q, grey=grey)
    data = J.synth.encode_jpeg(img, "4:4:4" if grey else mode, q, 0, optimise=opt)
    std = J.synth.encode_jpeg(img, "4:4:4" if grey else mode, q, 0)
    if opt:
        assert len(data) < len(std)
    a, b, c = ref.decode(data), oracle.decode(data), oracle.decode(std)
    assert np.array_equal(a.coefs, b.coefs) and np.array_equal(a.rgb, b.rgb)
    assert np.array_equal(b.coefs, c.coefs) and np.array_equal(b.rgb, c.rgb)      # the tables change the bytes, not the picture


@pytest.mark.parametrize("sampling", ["4:4:4", "4:2:0"])
def test_oracle_equals_reference_where_green_wraps(oracle, ref, sampling):
    """A whole file whose decoded samples reach (219, 228, 28), the triple whose green the
    reference's uint8 store wraps to 0 (colour_conversion.c:71-79): both agree end to end."""
    from conftest import wrap_triple_pixels, ycc_neighbourhood_jpeg
    data = ycc_neighbourhood_jpeg(sampling)
    o, r = oracle.decode(data), ref.decode(data)
    assert np.array_equal(o.coefs, r.coefs) and np.array_equal(o.rgb, r.rgb)
    hit = wrap_triple_pixels(o.planes, sampling)
    assert hit.sum() >= 64 and set(r.rgb[hit][:, 1].tolist()) == {0}


@pytest.mark.parametrize("sampling,pattern,grey", [("4:2:0", [0, 0, 0, 0, 1, 2], False), ("4:4:4", [0], True), ("4:4:4", [0, 1, 2], False)],
                         ids=["420", "grey", "444"])
def test_tables_that_need_the_canonical_walk(oracle, ref, sampling, pattern, grey):
    """AC tables whose frequent symbols have 11-bit codes over 76 ten-bit prefixes -- more than
    the kernels keep second-level tables for, so most symbols are found by the canonical
    ranges (decode.c:674-681): reference, restatement and the emulated kernels agree."""
    import jpeg_decoder_amd as J
    import huffcraft
    src = J.synth.encode_jpeg(J.synth.synth_image(160, 96, seed=5, grey=grey), sampling, 92, 0)
    o = oracle.decode(src)
    data, tabs = huffcraft.recode(src, o.coefs, pattern)
    assert all(huffcraft.long_prefixes(t) > 8 for t in tabs.values())
    o2, r2 = oracle.decode(data), ref.decode(data)
    assert np.array_equal(o2.coefs, o.coefs) and np.array_equal(o2.rgb, o.rgb)
    assert np.array_equal(r2.coefs, o2.coefs) and np.array_equal(r2.rgb, o2.rgb)
    got, _ = J.Emu().decode_coefs(data, o.coefs.shape[0] + 16, 0, 0)
    assert np.array_equal(got[:o.coefs.shape[0]], o.coefs)


def test_tables_with_long_dc_codes_and_sixteen_bit_ac_codes(oracle, ref):
    """DC differences coded with 12-bit codes (second-level lookup in the kernels) and nearly
    all AC symbols with 16-bit codes: reference, restatement and emulated kernels agree."""
    import jpeg_decoder_amd as J
    import huffcraft
    src = J.synth.encode_jpeg(J.synth.synth_image(160, 96, seed=6), "4:2:0", 92, 0)
    o = oracle.decode(src)
    data, _ = huffcraft.recode(src, o.coefs, [0, 0, 0, 0, 1, 2], huffcraft.sixteen_bit_ac_table, {2: [0], 12: list(range(1, 12))})
    o2, r2 = oracle.decode(data), ref.decode(data)
    assert np.array_equal(o2.coefs, o.coefs) and np.array_equal(o2.rgb, o.rgb)
    assert np.array_equal(r2.coefs, o2.coefs) and np.array_equal(r2.rgb, o2.rgb)
    got, _ = J.Emu().decode_coefs(data, o.coefs.shape[0] + 16, 0, 0)
    assert np.array_equal(got[:o.coefs.shape[0]], o.coefs)


def _assert_dies_on_partly_consumed_ff(ref, oracle, data, cap_blocks):
    import signal
    t = ref.decode_traced(data, cap_blocks, cap_blocks)
    assert t.signal == signal.SIGSEGV and not t.finished
    want = oracle.decode(data, want_rgb=False).coefs
    n = t.coefs.shape[0]
    # every block the reference produced before it died equals the restatement's
    assert 0 < n < want.shape[0] and np.array_equal(t.coefs, want[:n])
    last = t.checks[-1]
    pos = int(last["pos"])
    # the check that went wrong: cursor inside a data FF, then its stuffed zero, then RSTn;
    # check_marker answers "restart" (bitstream.c:104-108) and next_byte_restart_marker skips
    # the stuffed pair only (bitstream.c:144-151), leaving the cursor on the marker
    assert last["ret"] == 1 and 1 <= last["off"] <= 7 and last["blocks"] == n
    assert data[pos] == 0xFF and data[pos + 1] == 0 and data[pos + 2] == 0xFF and 0xD0 <= data[pos + 3] <= 0xD7
    assert last["pos_after"] == pos + 2
    # no such pattern before that point: the earlier restarts all ended behind their markers
    for c in t.checks[:-1]:
        if c["pos_after"] >= 0:
            after = int(c["pos_after"])
            assert data[after - 2] == 0xFF and 0xD0 <= data[after - 1] <= 0xD7
    return t


@pytest.mark.parametrize("name", __import__("conftest").FATAL_FIXTURES)
def test_reference_dies_on_fatal_fixture_as_recorded(ref, oracle, name):
    """The committed record (tests/golden/<name>.npz) is what this build of the reference does:
    it is ended by a signal, and its coefficient tap and restart checks up to then are the
    ones on file."""
    data, gold = load_fixture(name)
    t = _assert_dies_on_partly_consumed_ff(ref, oracle, data, 4096)
    assert np.array_equal(t.coefs, gold["coefs"])
    assert t.checks.tobytes() == gold["checks"].tobytes()


def test_reference_dies_the_same_way_on_a_config2_file(ref, oracle):
    """BASELINE config 2 (1920x1080 4:4:4, one restart interval per MCU row): about a fifth of
    the generated files hold the pattern.  Seed 1001: 43,200 of 97,200 blocks before the end."""
    import jpeg_decoder_amd as J
    data = J.synth.config_jpeg("C2", 1001)
    t = _assert_dies_on_partly_consumed_ff(ref, oracle, data, 100000)
    assert t.coefs.shape[0] == 43200 and len(t.checks) == 14401
