"""GPU: the pipelined front end (mj_pipe_*) against the one-shot path and the golden
outputs: batches of different shapes pushed through a few slots, from pageable memory
(gathered in a staging buffer), from page-locked memory with the files back to back (one
copy of the whole stretch; images are then read at whatever byte alignment they land on)
and from page-locked memory in descending address order (one copy per image); every image
of every batch is checked."""
import hashlib

import numpy as np
import pytest

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import PARITY_FIXTURES, load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    c = J.Context(0)
    yield c
    c.close()


def _digest(b, i):
    return hashlib.sha256(b.coefs(i).tobytes()).hexdigest(), hashlib.sha256(b.rgb(i).tobytes()).hexdigest()


@pytest.mark.parametrize("pinned", [False, True, "descending"])
@pytest.mark.parametrize("depth", [2, 3])
def test_pipeline_of_mixed_batches_equals_golden(ctx, depth, pinned):
    """Batches that differ in size, geometry and tables go through `depth` recycled slots;
    each slot is read before it is recycled.  Bit-exact against the reference's outputs."""
    fixtures = [load_fixture(n) for n in PARITY_FIXTURES]
    rng = np.random.default_rng(depth * 2 + bool(pinned))
    pipe = ctx.pipe(depth, 0)
    inflight = []
    keep = []

    def check(entry):
        b, idx = entry
        b.sync()
        for i, k in enumerate(idx):
            g = fixtures[k][1]
            assert b.status(i) == 0
            assert np.array_equal(b.coefs(i), g["coefs"]), PARITY_FIXTURES[k]
            assert np.array_equal(b.rgb(i), g["rgb"]), PARITY_FIXTURES[k]

    for step in range(9):
        idx = list(rng.choice(len(fixtures), size=int(rng.integers(1, 12)), replace=True))
        files = [fixtures[k][0] for k in idx]
        src = binding.PinnedFiles(files, descending=(pinned == "descending")) if pinned else files
        keep.append(src)
        if len(inflight) == depth:          # the slot about to be recycled
            check(inflight.pop(0))
        inflight.append((pipe.submit(src), idx))
    for e in inflight:
        check(e)
    pipe.close()
    for k in keep:
        if pinned:
            k.close()


def test_pipeline_at_baseline_shapes_equals_one_shot(ctx):
    """C3- and C2-shaped batches alternate through the pipeline; every image's coefficient and
    RGB digests equal those of the same image decoded alone by mj_batch_create."""
    sets = [[J.synth.config_jpeg("C3", seed=s) for s in range(3)], [J.synth.config_jpeg("C2", seed=s) for s in range(4)],
            [J.synth.config_jpeg("C3", seed=7), J.synth.config_jpeg("C1", seed=2)]]
    want = []
    for files in sets:
        row = []
        for f in files:
            b = ctx.batch([f], 0)
            b.decode()
            row.append(_digest(b, 0))
            b.close()
        want.append(row)
    pinned = [binding.PinnedFiles(f) for f in sets]
    pipe = ctx.pipe(2, 0)
    order = [0, 1, 2, 1, 0, 2, 0]
    prev = None
    for k in order:
        cur = (pipe.submit(pinned[k]), k)
        if prev is not None:
            b, pk = prev
            for i in range(len(sets[pk])):
                assert _digest(b, i) == want[pk][i], (pk, i)
        prev = cur
    b, pk = prev
    for i in range(len(sets[pk])):
        assert _digest(b, i) == want[pk][i], (pk, i)
    pipe.close()
    for p in pinned:
        p.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_pipeline_with_files_of_one_scan_per_component(ctx, oracle, pinned):
    """Files coded in one scan per component expand into extra units behind the files of the
    batch: through the pipeline's slots, from pageable and from page-locked memory, between
    ordinary files; every file against the restatement."""
    from conftest import SCANS_PER_COMPONENT_CASES, scans_per_component_pair
    pairs = [scans_per_component_pair(c) for c in SCANS_PER_COMPONENT_CASES[:4]]
    batches = [[pairs[0][1], pairs[0][0], pairs[1][0]], [pairs[2][0]], [pairs[3][1], pairs[1][1], pairs[3][0], pairs[2][1]]]
    pipe = ctx.pipe(2, 0)
    keep, done = [], []
    for files in batches * 2:
        src = binding.PinnedFiles(files) if pinned else files
        keep.append(src)
        done.append((pipe.submit(src), files))
        if len(done) == 2:
            b, fl = done.pop(0)
            b.sync()
            for i, data in enumerate(fl):
                o = oracle.decode(data)
                assert b.status(i) == 0
                assert np.array_equal(b.coefs(i), o.coefs) and np.array_equal(b.rgb(i), o.rgb)
    for b, fl in done:
        b.sync()
        for i, data in enumerate(fl):
            o = oracle.decode(data)
            assert np.array_equal(b.coefs(i), o.coefs) and np.array_equal(b.rgb(i), o.rgb)
    pipe.close()


def test_unknown_flag_bits_are_refused(ctx):
    data, _ = load_fixture("c420_q85_64x48")
    with pytest.raises(RuntimeError, match="unknown flag"):
        ctx.batch([data], 0x10000)
    with pytest.raises(RuntimeError, match="unknown flag"):
        ctx.pipe(2, 0x40)


def test_destroying_a_pipeline_batch_does_nothing(ctx):
    """A batch that mj_pipe_submit returned belongs to its pipeline: mj_batch_destroy leaves it
    alone, and the slot is good for the next submits and for mj_pipe_destroy."""
    data, g = load_fixture("c420_q85_100x75_ri3")
    pipe = ctx.pipe(2, 0)
    b = pipe.submit([data])
    b.sync()
    ctx.lib.mj_batch_destroy(b.h)                 # ignored
    assert np.array_equal(b.rgb(0), g["rgb"])     # still there
    for _ in range(4):                            # both slots used again, twice
        b = pipe.submit([data, data])
        assert np.array_equal(b.coefs(1), g["coefs"]) and np.array_equal(b.rgb(1), g["rgb"])
        ctx.lib.mj_batch_destroy(b.h)
    pipe.close()
