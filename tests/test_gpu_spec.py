"""GPU parity of the speculative synchronisation (mj_k_spec_*): the same bits as the rounds,
whatever its parameters, checked against the committed reference outputs and the CPU
restatement; and the library's own choice between the two."""
import numpy as np
import pytest

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import PARITY_FIXTURES, SCAN_GROUP_CASES, load_fixture, scan_group_pair

pytestmark = pytest.mark.gpu

# (sub_bits, lane_subs, extra): the defaults; lanes of two; short lanes with one extra subsequence, where
# the chain guesses all the time and the mending does the work; long lanes; odd subsequence lengths
FORMS = [(0, 0, 0), (256, 2, 3), (64, 1, 1), (96, 2, 1), (128, 8, 4), (512, 4, 2), (1024, 1, 2)]


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    c = J.Context(0)
    yield c
    c.close()


@pytest.fixture()
def policy():
    """Sets the process-wide policy for one test and puts the default back."""
    yield binding.set_sync_policy
    binding.set_sync_policy(binding.SYNC_AUTO)


def _check(b, i, coefs, rgb, what):
    assert b.status(i) == 0, f"{what}: status {b.status(i)}"
    got = b.coefs(i)
    bad = np.nonzero((got != coefs).any(axis=1))[0]
    assert bad.size == 0, f"{what}: {bad.size} coefficient blocks differ, first {bad[:5]}"
    assert np.array_equal(b.rgb(i), rgb), f"{what}: RGB differs"


def _speculative(b):
    st = b.entropy_stats()
    return st["workgroups"] == 0 and st["subsequences"] > 0


@pytest.mark.parametrize("form", FORMS)
def test_golden_fixtures_one_mixed_batch_speculative(ctx, policy, form):
    policy(binding.SYNC_SPECULATIVE, *form)
    files, golds = zip(*(load_fixture(n) for n in PARITY_FIXTURES))
    b = ctx.batch(list(files), binding.MJ_WANT_PLANES)
    for _ in range(2):
        b.decode()
        b.sync()
        assert _speculative(b)
        for i, (name, g) in enumerate(zip(PARITY_FIXTURES, golds)):
            _check(b, i, g["coefs"], g["rgb"], f"{name} {form}")
            for c in range(3):
                if f"plane{c}" in g:
                    assert np.array_equal(b.plane(i, c), g[f"plane{c}"]), f"{name} {form}: plane {c}"
    b.close()


@pytest.mark.parametrize("form", FORMS)
def test_random_images_speculative_vs_oracle(ctx, oracle, policy, form):
    rng = np.random.default_rng(11)
    samplings = ["4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1"]
    files, meta = [], []
    for _ in range(20):
        grey = rng.random() < 0.15
        w, h = int(rng.integers(1, 900)), int(rng.integers(1, 600))
        if grey:
            w = max(8, w // 8 * 8)
        samp = samplings[int(rng.integers(0, len(samplings)))]
        q = int(rng.integers(15, 101))
        ri = int(rng.choice([0, 0, 0, 1, 3, 16, 40]))
        img = J.synth.synth_image(w, h, seed=int(rng.integers(0, 1 << 30)), sigma=float(rng.choice([0.0, 2.0, 12.0, 40.0])), grey=grey)
        files.append(J.synth.encode_jpeg(img, samp, q, ri, optimise=bool(rng.random() < 0.3)))
        meta.append((w, h, samp, q, ri, grey))
    wants = [oracle.decode(d) for d in files]
    policy(binding.SYNC_SPECULATIVE, *form)
    b = ctx.batch(files, 0)
    b.decode()
    assert _speculative(b)
    for i, o in enumerate(wants):
        _check(b, i, o.coefs, o.rgb, f"{meta[i]} {form}")
    b.close()


def test_config1_and_a_long_scan_speculative_and_by_rounds(ctx, oracle, policy):
    """BASELINE config 1 (512 x 512 4:2:0, no restart markers) alone and eight at a time, and a
    1600 x 1200 4:2:2 file without restart markers (its one interval spans many sequences):
    rounds and speculation give the bits of the restatement; left to itself the library
    speculates on these and does not on a file with a restart interval per MCU row."""
    c1 = [J.synth.config_jpeg("C1", 1000 + i) for i in range(8)]
    big = J.synth.encode_jpeg(J.synth.synth_image(1600, 1200, seed=5, sigma=8.0), "4:2:2", 90, 0)
    rows = J.synth.config_jpeg("C2", 1000)
    want = {id(d): oracle.decode(d) for d in c1 + [big, rows]}
    for mode, files, expect in [(binding.SYNC_ROUNDS, c1[:1], False), (binding.SYNC_SPECULATIVE, c1[:1], True),
                                (binding.SYNC_AUTO, c1[:1], True), (binding.SYNC_AUTO, c1, True),
                                (binding.SYNC_ROUNDS, [big], False), (binding.SYNC_SPECULATIVE, [big], True),
                                (binding.SYNC_AUTO, [rows], False), (binding.SYNC_SPECULATIVE, [rows], True),
                                (binding.SYNC_SPECULATIVE, c1 + [big, rows], True)]:
        policy(mode)
        b = ctx.batch(files, 0)
        b.decode()
        assert _speculative(b) == expect, (mode, len(files))
        for i, d in enumerate(files):
            _check(b, i, want[id(d)].coefs, want[id(d)].rgb, f"mode {mode} file {i} of {len(files)}")
        b.close()


def test_subsequence_size_follows_bits_per_block(ctx, oracle, policy):
    """Left to itself the library cuts a speculating file into subsequences of 256 bits, or of the
    next power of two that holds two blocks of the file: a 512 x 512 4:2:0 file of some ninety bits
    per block keeps 256 (a lane twice as long takes twice as long), a 4:2:0 file at q 95 on a
    noisy picture, near 190 bits per block, gets 512.  A file or two on their own get shorter
    ones still: two blocks rounded up to 32 bits, 160 at least."""
    policy(binding.SYNC_AUTO)
    light = [J.synth.config_jpeg("C1", 1001 + i) for i in range(8)]
    heavy = J.synth.encode_jpeg(J.synth.synth_image(640, 480, seed=7, sigma=12.0), "4:2:0", 95, 0)
    for files, blocks, rule in [(light, 1024 * 6, "coarse"), (light[:1], 1024 * 6, "fine"), ([heavy], 1200 * 6, "coarse")]:
        b = ctx.batch(files, 0)
        b.decode()
        assert _speculative(b)
        want_subs = 0
        for i in range(len(files)):
            bits = b.info(i).entropy_bytes * 8            # with the stuffed bytes, which is what the library goes by
            two_blocks = 2 * (bits // blocks)
            if rule == "fine" and two_blocks <= 256:
                want_bits = max(160, (two_blocks + 31) // 32 * 32)
            else:
                want_bits = 256
                while want_bits < two_blocks:
                    want_bits *= 2
            if files is light or rule == "fine":
                assert want_bits == (256 if rule == "coarse" else 192), (bits, want_bits)
            else:
                assert want_bits == 512, (bits, want_bits)
            want_subs += bits / want_bits
        subs = b.entropy_stats()["subsequences"]
        assert want_subs * 0.97 < subs <= want_subs + len(files), (subs, want_subs, rule)
        for i, d in enumerate(files[:2]):
            o = oracle.decode(d)
            _check(b, i, o.coefs, o.rgb, f"{rule} subsequences, file {i}")
        b.close()


def test_files_of_many_bits_per_block(ctx, oracle, policy):
    """4:4:4 at q 98 on a noisy picture, 280 bits per block: many blocks fill up without an
    end-of-block symbol and the variants of a lane stay out of step, so speculation guesses at
    several per cent of the lane boundaries.  Forced, it still gives the restatement's bits;
    left to itself the library takes the rounds for such a file."""
    d = J.synth.encode_jpeg(J.synth.synth_image(640, 480, seed=7, sigma=12.0), "4:4:4", 98, 0)
    o = oracle.decode(d)
    for mode, expect in [(binding.SYNC_AUTO, False), (binding.SYNC_SPECULATIVE, True)]:
        policy(mode)
        b = ctx.batch([d], 0)
        b.decode()
        st = b.entropy_stats()
        assert _speculative(b) == expect
        if expect:
            assert st["rounds_sum"] > 100          # hops that were guesses
        _check(b, 0, o.coefs, o.rgb, f"dense file, mode {mode}")
        b.close()


@pytest.mark.parametrize("form", [(256, 2, 3), (128, 1, 3), (64, 1, 1)])
def test_several_intervals_of_many_sequences(ctx, oracle, policy, form):
    """A 2048 x 2048 file with a restart interval of 8192 MCUs: two intervals of a hundred
    sequences each with the default lanes (more than a wavefront folds by itself: the carry
    kernel takes every long interval of the list), beside a small file in the same batch."""
    big = J.synth.encode_jpeg(J.synth.synth_image(2048, 2048, seed=21, sigma=6.0), "4:2:0", 90, 8192)
    small = J.synth.config_jpeg("C1", 1001)
    wants = [oracle.decode(big), oracle.decode(small)]
    policy(binding.SYNC_SPECULATIVE, *form)
    b = ctx.batch([big, small], 0)
    for _ in range(2):
        b.decode()
        assert _speculative(b)
        st = b.entropy_stats()
        assert st["multi_seq_intervals"] == 3 and st["sequences"] > 2 * 64 + 17
        for i, o in enumerate(wants):
            _check(b, i, o.coefs, o.rgb, f"file {i} {form}")
    b.close()


def test_guesses_are_mended(ctx, oracle, policy):
    """Lanes of one 64-bit subsequence and one more: the chain has to guess hundreds of times in
    a 512 x 512 file; the counters say so and the result is the restatement's."""
    d = J.synth.config_jpeg("C1", 1003)
    o = oracle.decode(d)
    policy(binding.SYNC_SPECULATIVE, 64, 1, 1)
    b = ctx.batch([d], 0)
    b.decode()
    st = b.entropy_stats()
    assert st["workgroups"] == 0 and st["rounds_sum"] > 100 and st["round_tasks"] >= st["rounds_sum"]
    _check(b, 0, o.coefs, o.rgb, "guessed and mended")
    b.close()


@pytest.mark.parametrize("case", SCAN_GROUP_CASES[:2])
def test_files_of_several_scans_speculative(ctx, oracle, policy, case):
    multi, twin = scan_group_pair(case)
    policy(binding.SYNC_SPECULATIVE)
    b = ctx.batch([multi, twin], 0)
    b.decode()
    assert np.array_equal(b.rgb(0), b.rgb(1)), case
    assert np.array_equal(b.rgb(1), oracle.decode(twin).rgb), case
    b.close()


def test_damaged_streams_speculative(ctx, oracle, policy):
    """Damage inside the entropy-coded segment under speculation: no fault, the intact files of
    the batch decode as ever, a cut stream says so in its status word."""
    rng = np.random.default_rng(3)
    good = [J.synth.config_jpeg("C1", 1000), J.synth.encode_jpeg(J.synth.synth_image(320, 240, seed=2, sigma=6.0), "4:2:0", 80, 5)]
    wants = [oracle.decode(d) for d in good]
    for form in [(0, 0, 0), (64, 1, 1), (128, 8, 4)]:
        policy(binding.SYNC_SPECULATIVE, *form)
        files = list(good)
        for k in range(12):
            d = bytearray(good[k % 2])
            sos = d.index(b"\xff\xda")
            lo = sos + 14
            if k % 3 == 0:
                del d[lo + int(rng.integers(0, len(d) - lo - 2)):]       # cut
                d += b"\xff\xd9"
            elif k % 3 == 1:
                for _ in range(4):
                    d[lo + int(rng.integers(0, len(d) - lo - 2))] ^= 1 << int(rng.integers(0, 8))
            else:
                for _ in range(3):
                    p = lo + int(rng.integers(0, len(d) - lo - 3))
                    d[p] = 0xFF
                    d[p + 1] = 0xD0 + int(rng.integers(0, 8))
            files.append(bytes(d))
        b = ctx.batch(files, 0)
        b.decode()
        b.sync()
        for i, o in enumerate(wants):
            _check(b, i, o.coefs, o.rgb, f"intact file {i} beside damaged ones, {form}")
        assert any(b.status(i) != 0 for i in range(2, len(files)))
        b.close()


def test_policy_arguments_are_checked():
    lib = J.load_library()
    for bad in [(3, 0, 0, 0), (-1, 0, 0, 0), (2, 0, 3, 0), (2, 0, 16, 0), (2, 0, 1, 9), (2, 40000, 1, 1)]:
        assert lib.mj_set_sync_policy(*bad) == -1, bad
    assert lib.mj_set_sync_policy(0, 0, 0, 0) == 0
