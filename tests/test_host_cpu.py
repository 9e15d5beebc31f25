"""CPU-side checks: the C-ABI library loads and exports what include/*.h declares,
fails loudly without a GPU, and the host logic (parsers, table flattening, the
entropy algorithm in emulation) reproduces the reference's coefficients."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import ALL_FIXTURES, FATAL_FIXTURES, PARITY_FIXTURES, ROOT, load_fixture


def test_library_exports_every_declared_symbol():
    lib = J.load_library()
    for name in binding.ABI_SYMBOLS + binding.COMPAT_SYMBOLS:
        assert hasattr(lib, name), f"libmi355jpeg.so does not export {name}"


def test_symbol_lists_cover_the_headers():
    """Every function declared in include/mi355jpeg*.h is in the lists above."""
    declared = set()
    for h in ("mi355jpeg.h", "mi355jpeg_compat.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text):
            declared.add(m.group(1))
    declared -= {"defined"}
    missing = declared - set(binding.ABI_SYMBOLS + binding.COMPAT_SYMBOLS)
    assert not missing, f"declared but not listed/exported: {sorted(missing)}"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        J.Context(0)


def test_dropin_entry_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = J.load_library()
    data, _ = load_fixture("c420_q85_64x48")
    fh = (C.c_uint8 * 64)()
    C.cast(fh, C.POINTER(C.c_int32))[0] = -1
    img = C.c_void_p()
    buf = C.create_string_buffer(data, len(data) + 16)
    lib.decode_jpeg_buffer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.decode_jpeg_buffer(buf, len(data), fh, C.byref(img)) == -1
    assert not img.value


@pytest.mark.parametrize("name", ALL_FIXTURES + FATAL_FIXTURES)
@pytest.mark.parametrize("sub_bits,lane_subs", [(0, 0), (32, 1), (32, 3), (96, 2), (64, 8), (1024, 1), (512, 4)])
def test_entropy_algorithm_emulated(name, sub_bits, lane_subs):
    """Host parsers + table flattening + scan/unstuff rules + subsequence
    synchronisation, prefix sums and write pass (the kernels' algorithm run lane by
    lane on the CPU) against the reference's coefficient blocks."""
    emu = J.Emu()
    data, gold = load_fixture(name)
    nref = gold["coefs"].shape[0]
    got, st = emu.decode_coefs(data, nref + 16, sub_bits, lane_subs)
    assert got.shape[0] >= nref
    assert np.array_equal(got[:nref], gold["coefs"])
    if sub_bits:
        assert st["sub_bits"] == (sub_bits + 31) // 32 * 32
        assert st["lane_subs"] == lane_subs


@pytest.mark.parametrize("name", ALL_FIXTURES + FATAL_FIXTURES)
@pytest.mark.parametrize("sub_bits,lane_subs,extra", [(256, 1, 3), (256, 2, 3), (32, 1, 1), (64, 2, 1), (96, 4, 2), (128, 8, 4), (512, 1, 2)])
def test_speculative_synchronisation_emulated(name, sub_bits, lane_subs, extra):
    """The records looked up instead of decoded (every lane once per block slot, the chain of
    candidates through the maps, mending where it had to guess): the same coefficient blocks as
    the reference's, whatever the lanes are; short lanes with one extra subsequence make the
    chain guess all the time."""
    emu = J.Emu()
    data, gold = load_fixture(name)
    nref = gold["coefs"].shape[0]
    got, st = emu.decode_coefs(data, nref + 16, sub_bits, lane_subs, extra)
    assert np.array_equal(got[:nref], gold["coefs"])
    assert st["rounds_sum"] == 0 and st["fix_iterations"] == 0       # no rounds were run
    assert st["spec_mended"] >= st["spec_guesses"]


def test_speculative_synchronisation_guesses_and_mends():
    """On the 512 x 512 file: three extra subsequences per lane of one never guess; one extra
    subsequence of 64 bits guesses often, and the mending decodes from every guess."""
    emu = J.Emu()
    data, gold = load_fixture("C1_c420_q85_512x512")
    got, st = emu.decode_coefs(data, gold["coefs"].shape[0], 256, 1, 3)
    assert np.array_equal(got, gold["coefs"]) and st["spec_guesses"] == 0
    got, st = emu.decode_coefs(data, gold["coefs"].shape[0], 64, 1, 1)
    assert np.array_equal(got, gold["coefs"]) and st["spec_guesses"] > 100 and st["spec_mended_max"] > 1


def test_speculative_map_primitives():
    """mj_spec_core.h through the emulation's export: indices of the candidates, who goes on,
    who ends, and the composition of maps."""
    lib = C.CDLL(binding.EMU_PATH)
    lib.mj_emu_spec_selftest.restype = C.c_int
    assert lib.mj_emu_spec_selftest() == 0


def test_emulation_exercises_the_cross_sequence_path():
    emu = J.Emu()
    data, gold = load_fixture("C1_c420_q85_512x512")
    got, st = emu.decode_coefs(data, gold["coefs"].shape[0], 128, 1)
    assert st["multi_seq_intervals"] == 1 and st["sequences"] > 8 and st["fix_iterations"] >= 1
    assert np.array_equal(got, gold["coefs"])


def test_synthetic_generator_is_decodable_and_deterministic(oracle):
    img = J.synth.synth_image(96, 64, 5)
    for samp, ri in (("4:2:0", 0), ("4:2:0", 6), ("4:4:4", 12), ("4:2:2", 3), ("4:4:0", 0), ("4:1:1", 2)):
        a = J.synth.encode_jpeg(img, samp, 85, ri)
        b = J.synth.encode_jpeg(img, samp, 85, ri)
        assert a == b
        d = oracle.decode(a)
        assert (d.X, d.Y, d.ncs) == (96, 64, 3)
        assert np.abs(d.rgb.astype(int) - img).mean() < 12.0
        got, _ = J.Emu().decode_coefs(a, d.coefs.shape[0], 64, 2)
        assert np.array_equal(got, d.coefs)
    g = J.synth.encode_jpeg(img[..., 1].copy(), quality=75, restart_interval=4)
    d = oracle.decode(g)
    assert d.ncs == 1 and np.abs(d.rgb[..., 0].astype(int) - img[..., 1]).mean() < 8.0


def test_sixteen_byte_classifier_equals_the_byte_rule():
    """The scan kernels classify sixteen bytes at a time with byte masks; the result must be
    the byte-at-a-time rule (bitstream.c:22-59, :84-134, :142-176) on marker-heavy data,
    including segments that end inside a group of sixteen with other data behind them."""
    lib = C.CDLL(binding.EMU_PATH)
    lib.mj_emu_classify16.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.mj_emu_classify.restype = C.c_int
    rng = np.random.default_rng(7)
    alphabet = np.array([0xFF, 0xFF, 0xFF, 0x00, 0x00, 0xD0, 0xD3, 0xD7, 0xD8, 0xD9, 0xCF, 0x12, 0x80, 0xFE, 0x01],
                        np.uint8)
    out = np.zeros(16, np.uint8)
    for _ in range(600):
        n = int(rng.integers(1, 120))
        buf = rng.integers(0, 256, ((n + 15) // 16) * 16 + 32).astype(np.uint8)
        buf[:n] = alphabet[rng.integers(0, len(alphabet), n)]
        for pos in range(0, n, 16):
            lib.mj_emu_classify16(buf.ctypes.data, n, pos, out.ctypes.data)
            for i in range(16):
                j = pos + i
                if j >= n:
                    want = 1   # MJ_B_DROP
                else:
                    prev = int(buf[j - 1]) if j else 0
                    nxt = int(buf[j + 1]) if j + 1 < n else 0
                    want = lib.mj_emu_classify(prev, int(buf[j]), nxt)
                assert want == out[i], (n, j, hex(int(buf[j])))


def test_damaged_headers_and_streams_are_rejected_or_decoded_never_crash():
    """Random damage to the marker segments and to the entropy-coded data, through the host
    parser and the host emulation of the kernels' algorithm: every call returns (the same
    loop runs clean under AddressSanitizer and UBSan when the sources are built with
    them)."""
    emu = binding.Emu()
    rng = np.random.default_rng(11)
    done = rejected = 0
    for name in ["c420_q85_100x75_ri3", "g_q75_72x40_ri4", "c444_q90_dqt16_56x40"]:
        good, _ = load_fixture(name)
        sos = good.index(b"\xff\xda")
        hdr = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3])
        for v in range(150):
            d = bytearray(good)
            kind = v % 5
            if kind == 0:
                d[2 + int(rng.integers(0, hdr - 2))] = int(rng.integers(0, 256))
            elif kind == 1:
                d = d[:2 + int(rng.integers(0, hdr))]
            elif kind == 2:
                p = 2 + int(rng.integers(0, hdr - 4))
                d[p], d[p + 1] = 0xFF, 0xC0 + int(rng.integers(0, 0x3F))
            elif kind == 3:
                d = d[:hdr + int(rng.integers(0, len(good) - hdr))]
            else:
                d[hdr + int(rng.integers(0, len(good) - hdr - 2))] ^= 1 << int(rng.integers(0, 8))
            try:
                emu.decode_coefs(bytes(d), 4096)
                done += 1
            except RuntimeError:
                rejected += 1
    assert done + rejected == 450 and done > 0 and rejected > 0


def test_host_code_is_clean_under_sanitizers(tmp_path):
    """The host parser, the table flattening and the emulated kernel algorithm, compiled with
    AddressSanitizer + UBSan (leak check included) and fed a few thousand damaged files."""
    import shutil
    import subprocess
    if not (shutil.which("gcc") and shutil.which("g++")):
        pytest.skip("no host compiler")
    pkg = os.path.join(ROOT, "jpeg-decoder_amd")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"), "-I" + os.path.join(pkg, "host")]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    objs = []
    srcs = [os.path.join(pkg, "host", n + ".c") for n in ("bitcursor", "segments", "flatten", "decode_host")]
    srcs.append(os.path.join(ROOT, "tests", "fuzz", "stub_abi.c"))      # decode_host.c without a device behind it
    for src in srcs:
        obj = str(tmp_path / (os.path.basename(src)[:-2] + ".o"))
        subprocess.run(["gcc", "-std=c17", "-D_POSIX_C_SOURCE=200809L", *san, *inc, "-c", src, "-o", obj], check=True)
        objs.append(obj)
    exe = str(tmp_path / "fuzz_host")
    subprocess.run(["g++", "-std=c++17", *san, *inc, "-o", exe, os.path.join(ROOT, "tests", "fuzz", "fuzz_host.cpp"),
                    os.path.join(pkg, "csrc", "mj_emu.cpp"), *objs, "-lm", "-lpthread"], check=True)
    files = [os.path.join(ROOT, "tests", "golden", n + ".jpg")
             for n in ("c420_q85_100x75_ri3", "g_q75_72x40_ri4", "c422_q95_opt_136x40_ri5", "c444_q90_dqt16_56x40")]
    import jpeg_decoder_amd as J
    scans = str(tmp_path / "one_scan_per_component.jpg")      # three scans: the walk goes on after each
    with open(scans, "wb") as f:
        f.write(J.synth.encode_jpeg(J.synth.synth_image(100, 75, 3), "4:2:0", 85, 9, scans_per_component=True))
    files.append(scans)
    r = subprocess.run([exe, "1100", *files], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runs 5500" in r.stdout


def test_reference_viewer_translation_unit_links_against_the_library(tmp_path):
    """A translation unit with the include list of the reference's viewer
    (jpeg_decoder.c:14-21: colour_conversion.h, debug.h, decode.h, frame_header.h,
    huff_table.h, restart_interval.h, scan_header.h) and its call sequence
    (jpeg_decoder.c:52-104: decode_jpeg_buffer, y_rgb / ycbcr_rgb / yccb_rgb, free_img,
    free_frame_header, debug_print) compiles against include/ and links against
    libmi355jpeg.so alone."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no host compiler")
    pkg = os.path.join(ROOT, "jpeg-decoder_amd")
    src = os.path.join(pkg, "host", "jpeg_decoder_main.c")
    text = open(src).read()
    for h in ("colour_conversion.h", "debug.h", "decode.h", "frame_header.h", "huff_table.h", "restart_interval.h", "scan_header.h"):
        assert f'#include "{h}"' in text and os.path.exists(os.path.join(ROOT, "include", h)), h
    for call in ("decode_jpeg_buffer(", "y_rgb(", "ycbcr_rgb(", "yccb_rgb(", "free_img(", "free_frame_header(", "debug_print("):
        assert call in text, call
    exe = str(tmp_path / "jpeg_decoder")
    subprocess.run(["gcc", "-std=c17", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-I" + os.path.join(ROOT, "include"),
                    "-o", exe, src, "-L" + pkg, "-lmi355jpeg", "-Wl,-rpath," + pkg], check=True)
    import torch
    if not torch.cuda.is_available():
        # without a device the executable must fail loudly, not print a checksum
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "C1_c420_q85_512x512.jpg")], capture_output=True, text=True)
        assert r.returncode != 0 and "fnv1a" not in r.stdout


def test_product_library_does_not_contain_or_need_the_checkers():
    """libmi355jpeg.so exports no symbol of the CPU restatement (orc_*) or of the compiled
    reference's harness (ref_*), and depends on no library under oracle/."""
    import shutil
    import subprocess
    if not (shutil.which("nm") and shutil.which("ldd")):
        pytest.skip("no binutils")
    syms = subprocess.run(["nm", "-D", binding.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in syms.splitlines() if ln.strip()]
    bad = [n for n in names if n.startswith(("orc_", "ref_", "__wrap_", "__real_"))]
    assert not bad, bad
    deps = subprocess.run(["ldd", binding.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "oracle" not in deps and "liboracle" not in deps and "libjpegref" not in deps, deps
    for root, _, files in os.walk(os.path.join(ROOT, "jpeg-decoder_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".cpp", ".py")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "libjpegref" not in text and "refbind" not in text, os.path.join(root, f)
