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
