import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["fixtures"]


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
        data = f.read()
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return data, {k: z[k] for k in z.files}


PARITY_FIXTURES = [f["name"] for f in load_manifest() if not f["divergent"]]
ALL_FIXTURES = [f["name"] for f in load_manifest() if not f.get("fatal")]
FATAL_FIXTURES = [f["name"] for f in load_manifest() if f.get("fatal")]     # the compiled reference dies on these


@pytest.fixture(scope="session")
def oracle():
    from refbind import Oracle
    if not Oracle.available():
        pytest.skip("oracle/liboracle.so not built (run __graft_entry__.build())")
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from refbind import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libjpegref.so absent (built only where /root/reference exists)")
    return Ref()


def ycc_neighbourhood_jpeg(sampling):
    """A quality-100 file of flat MCUs whose (Y, Cb, Cr) sweep the neighbourhood of
    (219, 228, 28): the one triple whose green sum is exactly -256, which the reference's
    uint8 store turns into 0 instead of clamping (colour_conversion.c:71-79).  Returns the file."""
    import numpy as np
    import jpeg_decoder_amd as J
    m = 8 if sampling == "4:4:4" else 16
    combos = [(y, cb, cr) for y in range(214, 226) for cb in range(225, 232) for cr in range(25, 32)]
    cols = 28
    rows = (len(combos) + cols - 1) // cols
    img = np.zeros((rows * m, cols * m, 3), np.uint8)
    for i, t in enumerate(combos):
        r, c = divmod(i, cols)
        img[r * m:(r + 1) * m, c * m:(c + 1) * m] = t
    return J.synth.encode_jpeg(img, sampling, 100, 0, ycc=True)


def wrap_triple_pixels(planes, sampling):
    """Boolean mask of the luma plane where the decoded samples are (219, 228, 28)."""
    import numpy as np
    cb, cr = planes[1], planes[2]
    if sampling != "4:4:4":
        cb = np.repeat(np.repeat(cb, 2, 0), 2, 1)
        cr = np.repeat(np.repeat(cr, 2, 0), 2, 1)
    return (planes[0] == 219) & (cb == 228) & (cr == 28)


SCANS_PER_COMPONENT_CASES = [
    # (w, h, sampling, quality, restart interval, fitted tables)
    (96, 64, "4:2:0", 85, 0, False),        # whole MCUs
    (100, 75, "4:2:0", 85, 0, True),        # ragged edges: the scans of a component code fewer blocks than the MCU grid has
    (203, 77, "4:2:2", 90, 7, False),       # restart intervals counted in blocks
    (131, 77, "4:4:4", 75, 5, False),
    (64, 48, "4:1:1", 60, 0, True),
    (50, 70, "4:4:0", 95, 3, False),
    (331, 316, "4:4:0", 86, 3, True),       # fitted tables: the DC differences of a component's own raster order
    (29, 305, "4:1:1", 36, 7, True),
    (615, 25, "4:4:0", 80, 0, True),
]


def scans_per_component_pair(case):
    """(file with one scan per component, the same coefficients in one interleaved scan)."""
    import jpeg_decoder_amd as J
    w, h, sampling, q, ri, fitted = case
    img = J.synth.synth_image(w, h, seed=w * 3 + h)
    return (J.synth.encode_jpeg(img, sampling, q, ri, optimise=fitted, scans_per_component=True),
            J.synth.encode_jpeg(img, sampling, q, ri, optimise=fitted))


SCAN_GROUP_CASES = [
    # (w, h, sampling, quality, restart interval, fitted tables, how the components are dealt out over two scans)
    (96, 64, "4:2:0", 85, 0, False, "y|cbcr"),
    (100, 75, "4:2:0", 85, 0, True, "y|cbcr"),        # ragged edges: the luma scan codes fewer blocks than the MCU grid has
    (100, 75, "4:2:0", 85, 5, True, "ycb|cr"),
    (203, 77, "4:2:2", 90, 7, False, "y|cbcr"),       # restart intervals counted in each scan's own MCUs
    (131, 77, "4:4:4", 75, 5, False, "ycb|cr"),
    (64, 48, "4:1:1", 60, 0, True, "ycb|cr"),
    (50, 70, "4:4:0", 95, 3, False, "y|cbcr"),
    (331, 316, "4:2:0", 86, 11, True, "y|cbcr"),
    (615, 25, "4:4:0", 80, 0, True, "ycb|cr"),
]


def scan_group_pair(case):
    """(file whose frame takes two scans -- one component alone, two interleaved --, the same
    coefficients in one interleaved scan)."""
    import jpeg_decoder_amd as J
    w, h, sampling, q, ri, fitted, scans = case
    img = J.synth.synth_image(w, h, seed=w * 5 + h)
    return (J.synth.encode_jpeg(img, sampling, q, ri, optimise=fitted, scans=scans),
            J.synth.encode_jpeg(img, sampling, q, ri and 3, optimise=fitted))
