"""ctypes binding of libffs_synth.so (include/ffs_synth.h): deterministic
synthetic detector frames and masks for the SURVEY section 8(d) workloads."""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_LIB_PATH = os.path.join(_PKG, "libffs_synth.so")


class SynthParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixel_bytes", C.c_int32),
                ("seed", C.c_uint64), ("background", C.c_double), ("n_spots", C.c_uint32),
                ("sigma_min", C.c_double), ("sigma_max", C.c_double),
                ("peak_min", C.c_double), ("peak_max", C.c_double),
                ("max_value", C.c_uint32), ("n_frames", C.c_uint32),
                ("sigma_z_min", C.c_double), ("sigma_z_max", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `make synth` (or __graft_entry__.build())")
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def params(width, height, dtype=np.uint16, seed=0, background=2.0, n_spots=100,
           sigma=(0.8, 1.6), peak=(30.0, 5000.0), max_value=0, n_frames=0,
           sigma_z=(0.0, 0.0)) -> SynthParams:
    dt = np.dtype(dtype)
    assert dt in (np.dtype(np.uint16), np.dtype(np.uint32))
    return SynthParams(width, height, dt.itemsize, seed, background, n_spots,
                       sigma[0], sigma[1], peak[0], peak[1], max_value, n_frames,
                       sigma_z[0], sigma_z[1])


def frame(p: SynthParams, index: int, out: np.ndarray | None = None) -> np.ndarray:
    dt = np.uint16 if p.pixel_bytes == 2 else np.uint32
    if out is None:
        out = np.empty((p.height, p.width), dt)
    assert out.dtype == dt and out.shape == (p.height, p.width) and out.flags.c_contiguous
    rc = lib().ffs_synth_frame(C.byref(p), C.c_uint32(index), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("ffs_synth_frame rejected its parameters")
    return out


def frames(p: SynthParams, indices, threads: int = 8) -> np.ndarray:
    indices = list(indices)
    dt = np.uint16 if p.pixel_bytes == 2 else np.uint32
    out = np.empty((len(indices), p.height, p.width), dt)
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(lambda iz: frame(p, iz[1], out[iz[0]]), enumerate(indices)))
    return out


def mask_modules(width, height, mod_fast, mod_slow, gap_fast, gap_slow) -> np.ndarray:
    m = np.empty((height, width), np.uint8)
    lib().ffs_synth_mask_modules(m.ctypes.data_as(C.c_void_p), width, height,
                                 mod_fast, mod_slow, gap_fast, gap_slow)
    return m


def mask_eiger16m() -> np.ndarray:
    """Eiger 2XE 16M module-gap mask (h5read/include/eiger2xe.h:6-19)."""
    return mask_modules(4148, 4362, 1028, 512, 12, 38)


def mask_dead_pixels(mask: np.ndarray, seed: int, n_dead: int) -> np.ndarray:
    assert mask.dtype == np.uint8 and mask.flags.c_contiguous
    H, W = mask.shape
    lib().ffs_synth_mask_dead_pixels(mask.ctypes.data_as(C.c_void_p), W, H,
                                     C.c_uint64(seed), C.c_uint32(n_dead))
    return mask


def mask_rect(mask: np.ndarray, x0, x1, y0, y1) -> np.ndarray:
    H, W = mask.shape
    lib().ffs_synth_mask_rect(mask.ctypes.data_as(C.c_void_p), W, H, x0, x1, y0, y1)
    return mask


def reference_sample(n: int, dtype=np.uint16) -> np.ndarray:
    """The reference's generated sample image n (h5read.c:203-276), 4362 x 4148."""
    dt = np.dtype(dtype)
    out = np.empty((4362, 4148), dt)
    rc = lib().ffs_synth_reference_sample(C.c_uint32(n), C.c_int32(dt.itemsize),
                                          out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("bad sample index")
    return out


# ---- the workloads BASELINE.json names ---------------------

def config1_params(seed=1000) -> SynthParams:
    """10 x 1024^2 u16, lambda=3, 150 spots (plumbing config)."""
    return params(1024, 1024, np.uint16, seed=seed, background=3.0, n_spots=150,
                  sigma=(0.8, 1.6), peak=(30.0, 5000.0), max_value=65535)


def config1_mask() -> np.ndarray:
    m = np.ones((1024, 1024), np.uint8)
    mask_rect(m, 500, 512, 0, 1024)
    mask_rect(m, 0, 1024, 480, 518)
    return mask_dead_pixels(m, 1000, 50)


def eiger16m_params(seed=2000, background=2.0, n_spots=1500) -> SynthParams:
    """Eiger 2XE 16M u16 stills (config 2/3)."""
    return params(4148, 4362, np.uint16, seed=seed, background=background, n_spots=n_spots,
                  sigma=(0.8, 1.6), peak=(30.0, 5000.0), max_value=65535)


def jungfrau9m_params(seed=4000) -> SynthParams:
    """3072^2 u32, lambda=5, values < 2^24 (config 4)."""
    return params(3072, 3072, np.uint32, seed=seed, background=5.0, n_spots=1000,
                  sigma=(0.8, 1.6), peak=(30.0, 200000.0), max_value=(1 << 24) - 1)


def sweep_params(seed=5000, n_frames=100, n_spots=800, width=4148, height=4362) -> SynthParams:
    """Fine-phi sweep with rocking-curve spots (config 5)."""
    return params(width, height, np.uint16, seed=seed, background=2.0, n_spots=n_spots,
                  sigma=(0.8, 1.6), peak=(30.0, 5000.0), max_value=65535,
                  n_frames=n_frames, sigma_z=(0.5, 2.0))
