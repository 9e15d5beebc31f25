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
