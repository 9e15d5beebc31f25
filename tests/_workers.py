"""Worker functions for process pools in the tests (importable by name, so that a
`spawn` pool can run them: the GPU tests must not fork a process that holds a HIP
context)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def config_file_and_digests(args):
    """(file bytes, sha256 of the restatement's coefficients, sha256 of its RGB) for one
    image of a named BASELINE configuration."""
    config, seed = args
    import jpeg_decoder_amd as J
    from refbind import Oracle
    data = J.synth.config_jpeg(config, seed)
    want = Oracle().decode(data)
    return data, hashlib.sha256(want.coefs.tobytes()).hexdigest(), hashlib.sha256(want.rgb.tobytes()).hexdigest()


def pool_size():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))
