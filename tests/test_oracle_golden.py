"""The CPU restatement (oracle/) against the reference's committed outputs.

tests/golden/*.npz were produced by the compiled, unmodified reference
(oracle/gen_fixtures.py); the restatement must reproduce them bit for bit.
"""
import hashlib

import numpy as np
import pytest

from conftest import PARITY_FIXTURES, load_fixture, load_manifest


@pytest.mark.parametrize("name", PARITY_FIXTURES)
def test_oracle_matches_golden(oracle, name):
    data, gold = load_fixture(name)
    d = oracle.decode(data)
    assert d.coefs.shape == gold["coefs"].shape
    assert np.array_equal(d.coefs, gold["coefs"]), "Huffman-decoded coefficient blocks differ"
    for i, p in enumerate(d.planes):
        assert np.array_equal(p, gold[f"plane{i}"]), f"component plane {i} differs"
    assert np.array_equal(d.rgb, gold["rgb"]), "RGB differs"


def test_manifest_digests_match_files():
    for f in load_manifest():
        _, gold = load_fixture(f["name"])
        assert hashlib.sha256(np.ascontiguousarray(gold["coefs"]).tobytes()).hexdigest() == f["sha256_coefs"]
        assert hashlib.sha256(np.ascontiguousarray(gold["rgb"]).tobytes()).hexdigest() == f["sha256_rgb"]


def test_documented_divergence_last_mcu(oracle):
    """The reference ends a scan at the next marker, never by MCU count
    (decode.c:578-589), so a final MCU that begins inside the last entropy byte is
    lost.  The restatement decodes by count; the blocks both produce are equal."""
    data, gold = load_fixture("g_flat_32x8_refdrops")
    d = oracle.decode(data)
    assert gold["coefs"].shape[0] == 3 and d.coefs.shape[0] == 4
    assert np.array_equal(d.coefs[:3], gold["coefs"])


def test_integer_red_and_blue_equal_the_float_path(oracle):
    """The colour kernel computes R and B as y + floor(1.402 (cr - 128)) and
    y + floor(1.772 (cb - 128)), clamped, instead of the reference's double sum, float
    rounding and truncation (colour_conversion.c:71-79).  All 65,536 (y, chroma) pairs
    per channel must agree with the checker."""
    yy, cc = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    y, c = yy.ravel(), cc.ravel()
    mid = np.full_like(y, 128)

    def rows(a, b, d):
        return np.concatenate([oracle.ycbcr_row(a[i:i + 32768], b[i:i + 32768], d[i:i + 32768])
                               for i in range(0, a.size, 32768)])

    w = c.astype(int) - 128
    red = rows(y, mid, c)[:, 0].astype(int)
    blue = rows(y, c, mid)[:, 2].astype(int)
    assert np.array_equal(red, np.clip(y.astype(int) + np.floor_divide(701 * w, 500), 0, 255))
    assert np.array_equal(blue, np.clip(y.astype(int) + np.floor_divide(443 * w, 250), 0, 255))
    # and neither depends on the other chroma sample
    other = (c[::-1]).copy()
    assert np.array_equal(rows(y, other, c)[:, 0].astype(int), red)
    assert np.array_equal(rows(y, c, other)[:, 2].astype(int), blue)


def _case_id(c):
    return f"{c[0]}x{c[1]}_{c[2]}_q{c[3]}_ri{c[4]}" + ("_fitted" if c[5] else "")


from conftest import SCANS_PER_COMPONENT_CASES, scans_per_component_pair  # noqa: E402


@pytest.mark.parametrize("case", SCANS_PER_COMPONENT_CASES, ids=_case_id)
def test_one_scan_per_component_equals_the_interleaved_coding(oracle, case):
    """A frame coded in one scan per component (T.81 A.2.2).  The reference has no output for
    such files (it keeps counting MCUs across scans, decode.c:125, :473, :504, and writes past
    its planes), so the restatement is pinned through the interleaved coding of the same
    quantised coefficients, which the reference does decode: same RGB, same visible samples,
    and the same coefficient blocks once they are put in the same order."""
    multi, inter = scans_per_component_pair(case)
    assert multi.count(b"\xff\xda") == 3
    m, i = oracle.decode(multi), oracle.decode(inter)
    assert np.array_equal(m.rgb, i.rgb)
    w, h = case[0], case[1]
    hs, vs = m.hsf[0], m.vsf[0]
    assert np.array_equal(m.planes[0][:h, :w], i.planes[0][:h, :w])
    cw, ch = -(-w // hs), -(-h // vs)
    for c in (1, 2):
        assert np.array_equal(m.planes[c][:ch, :cw], i.planes[c][:ch, :cw])
    # coefficient blocks: component by component in raster order against MCU by MCU
    mcus_x = i.planes[0].shape[1] // (8 * hs)
    bpm = hs * vs + 2
    at = 0
    for c in range(3):
        hc, vc = (hs, vs) if c == 0 else (1, 1)
        base = 0 if c == 0 else hs * vs + c - 1
        bw, bh = m.planes[c].shape[1] // 8, m.planes[c].shape[0] // 8
        for by in range(bh):
            for bx in range(bw):
                src = ((by // vc) * mcus_x + bx // hc) * bpm + base + (by % vc) * hc + bx % hc
                assert np.array_equal(m.coefs[at], i.coefs[src]), (c, bx, by)
                at += 1
    assert at == m.coefs.shape[0]


from conftest import SCAN_GROUP_CASES, scan_group_pair  # noqa: E402


@pytest.mark.parametrize("case", SCAN_GROUP_CASES, ids=lambda c: _case_id(c) + "_" + c[6].replace("|", "_"))
def test_scans_of_some_of_the_components_equal_the_interleaved_coding(oracle, case):
    """A frame coded in two scans, one component alone and the other two interleaved (T.81 A.2.2,
    A.2.3).  PARITY UNPINNED against the reference, which has no output for files of several
    scans (decode.c:125, :473, :504); pinned, like one scan per component, through the
    interleaved coding of the same quantised coefficients: same RGB, same visible samples, and
    the same coefficient blocks once they are put in the same order."""
    multi, inter = scan_group_pair(case)
    assert multi.count(b"\xff\xda") == 2
    m, i = oracle.decode(multi), oracle.decode(inter)
    assert np.array_equal(m.rgb, i.rgb)
    w, h = case[0], case[1]
    hs, vs = m.hsf[0], m.vsf[0]
    assert np.array_equal(m.planes[0][:h, :w], i.planes[0][:h, :w])
    cw, ch = -(-w // hs), -(-h // vs)
    for c in (1, 2):
        assert np.array_equal(m.planes[c][:ch, :cw], i.planes[c][:ch, :cw])
    mcus_x, mcus_y = i.planes[0].shape[1] // (8 * hs), i.planes[0].shape[0] // (8 * vs)
    bpm = hs * vs + 2
    groups = [[0], [1, 2]] if case[6] == "y|cbcr" else [[0, 1], [2]]
    at = 0
    for grp in groups:
        if len(grp) == 1:           # the component's own blocks in raster order
            c = grp[0]
            hc, vc = (hs, vs) if c == 0 else (1, 1)
            base = 0 if c == 0 else hs * vs + c - 1
            bw, bh = m.planes[c].shape[1] // 8, m.planes[c].shape[0] // 8
            for by in range(bh):
                for bx in range(bw):
                    src = ((by // vc) * mcus_x + bx // hc) * bpm + base + (by % vc) * hc + bx % hc
                    assert np.array_equal(m.coefs[at], i.coefs[src]), (c, bx, by)
                    at += 1
        else:                       # the frame's MCUs with the blocks of the scan's components
            for mcu in range(mcus_x * mcus_y):
                for c in grp:
                    n, base = (hs * vs, 0) if c == 0 else (1, hs * vs + c - 1)
                    for k in range(n):
                        assert np.array_equal(m.coefs[at], i.coefs[mcu * bpm + base + k]), (c, mcu, k)
                        at += 1
    assert at == m.coefs.shape[0]


@pytest.mark.parametrize("name", __import__("conftest").FATAL_FIXTURES)
def test_fatal_fixture_prefix_equals_the_restatement(oracle, name):
    """Files the compiled reference dies on (its restart check takes a partly consumed data FF
    before a restart marker, FF 00 FF Dn, for the marker: bitstream.c:84-134, :142-151).  The
    committed record of what it did until then: every block it tapped equals the
    restatement's, and the place it went wrong is the byte pattern named."""
    from conftest import load_manifest
    data, gold = load_fixture(name)
    m = next(f for f in load_manifest() if f["name"] == name)
    d = oracle.decode(data)
    n = gold["coefs"].shape[0]
    assert 0 < n < d.coefs.shape[0] and n == m["ref_blocks"]
    assert np.array_equal(d.coefs[:n], gold["coefs"])
    ck = gold["checks"]
    last = ck[-1]
    # the last thing the reference did: a restart check that said "restart marker" ...
    assert last["ret"] == 1 and len(ck) - 1 == m["fatal_mcu"] and last["blocks"] == n
    pos = int(last["pos"])
    # ... with the cursor inside a data FF (1 to 7 bits of it consumed), its stuffed zero and the marker behind it
    assert 1 <= last["off"] <= 7 and data[pos] == 0xFF and data[pos + 1] == 0x00 and data[pos + 2] == 0xFF
    assert 0xD0 <= data[pos + 3] <= 0xD7 and data[pos:pos + 4].hex() == m["fatal_bytes"]
    # ... after which its cursor stood on the marker instead of behind it
    assert last["pos_after"] == pos + 2
    # every earlier restart was taken from a fully consumed byte and ended behind its marker
    for c in ck[:-1]:
        if c["pos_after"] >= 0:
            after = int(c["pos_after"])
            assert data[after - 2] == 0xFF and 0xD0 <= data[after - 1] <= 0xD7
