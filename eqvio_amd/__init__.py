"""eqvio_amd — MI355X-native (gfx950) EqF update path of EqVIO behind the reference's VIOFilter / VIO_eqf
interfaces. The compute path is hand-written HIP in eqvio_amd/csrc (libeqf_hip.so); this package is the
ctypes plumbing used by the tests and bench.py. No CPU fallback exists: without the HIP library and a
gfx950 device every entry point raises."""
from .capi import COORD_EUCLIDEAN, COORD_INVDEPTH, COORD_NORMAL, Camera, EqfCore, EqfError, Settings, VIOFilter, load_eqf_lib, load_filter_lib  # noqa: F401

__all__ = ["Camera", "EqfCore", "VIOFilter", "load_filter_lib", "EqfError", "Settings", "load_eqf_lib", "COORD_EUCLIDEAN", "COORD_INVDEPTH", "COORD_NORMAL"]
