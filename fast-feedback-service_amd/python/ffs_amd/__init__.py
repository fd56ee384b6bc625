"""ffs_amd -- Python mirror of libffs_hip.so's C ABI (include/ffs_hip.h).

The library is the product; this package is the thin host-side binding used by
tests/, bench.py and __graft_entry__.py.  It fails loudly when the HIP library is
missing -- there is no CPU fallback.
"""
from . import synth  # noqa: F401
from .api import (Context, FrameResult, Params, Stack3D, Stream, device_count,  # noqa: F401
                  device_name, lib_path, load_library, FfsError, BOX_DT, REFL_DT,
                  ALGO_DISPERSION, ALGO_DISPERSION_EXTENDED, multi_init)
