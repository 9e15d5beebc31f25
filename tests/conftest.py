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
ALL_FIXTURES = [f["name"] for f in load_manifest()]


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
