"""MI355X-native baseline JPEG decode path behind the API of Tommy-Sand/jpeg-decoder.

The product is ``libmi355jpeg.so`` (C host code + HIP kernels, C ABI declared in
``include/mi355jpeg.h`` and ``include/mi355jpeg_compat.h``).  This package is the
thin ctypes layer the tests and the benchmark drive it through.
"""
from . import shard, synth, topology  # noqa: F401
from .binding import Batch, Context, Emu, LIB_PATH, PinnedFiles, Pipe, load_library  # noqa: F401
