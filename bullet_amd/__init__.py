"""bullet_amd -- MI355X-native backend for the training hot path of jw1912/bullet.

The product is `libbullet_amd.so` (hand-written HIP kernels for gfx950 behind the C ABI of
include/bullet_amd.h).  This Python package is only plumbing for tests and bench.py: it loads the
library with ctypes and passes raw device pointers (PyTorch tensors are used purely as device
memory / streams).  There is NO CPU fallback: importing `bullet_amd.lib` without the built library
raises, and every op needs a GPU.
"""
from . import lib  # noqa: F401
from .lib import ACT, BtError, check, load  # noqa: F401
