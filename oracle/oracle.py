"""ctypes doors onto oracle/liboracle.so (our C restatement) and, when present,
oracle/_ref/libffs_ref.so (the reference's own standalone.cc compiled from source).

TEST INFRASTRUCTURE ONLY -- see oracle/ffs_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libffs_ref.so")


def build(quiet: bool = True) -> None:
    """Run oracle/Makefile (builds _ref only when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class DispParams(C.Structure):
    _fields_ = [("kernel_half_x", C.c_int), ("kernel_half_y", C.c_int),
                ("min_count", C.c_int), ("threshold", C.c_double),
                ("nsig_b", C.c_double), ("nsig_s", C.c_double)]


class Box(C.Structure):
    _fields_ = [("l", C.c_uint32), ("t", C.c_uint32), ("r", C.c_uint32),
                ("b", C.c_uint32), ("num_pixels", C.c_int32)]


class Reflection(C.Structure):
    _fields_ = [("x_min", C.c_uint32), ("x_max", C.c_uint32),
                ("y_min", C.c_uint32), ("y_max", C.c_uint32),
                ("z_min", C.c_int32), ("z_max", C.c_int32),
                ("num_pixels", C.c_int32),
                ("com_x", C.c_float), ("com_y", C.c_float), ("com_z", C.c_float),
                ("peak_x", C.c_uint32), ("peak_y", C.c_uint32), ("peak_z", C.c_int32),
                ("peak_intensity", C.c_uint32),
                ("peak_centroid_distance", C.c_float),
                ("sum_intensity", C.c_uint64)]


class Slice(C.Structure):
    _fields_ = [("n", C.c_size_t), ("linear_index", C.POINTER(C.c_uint64)),
                ("intensity", C.POINTER(C.c_uint32))]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.ffs_oracle_free.argtypes = [C.c_void_p]
    return _lib


def have_ref() -> bool:
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF)
        _ref.ffs_ref_create.restype = C.c_void_p
        _ref.ffs_ref_create.argtypes = [C.c_size_t, C.c_size_t]
        _ref.ffs_ref_destroy.argtypes = [C.c_void_p]
        _ref.ffs_ref_standard_dispersion.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    return _ref


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def dispersion(image: np.ndarray, mask: np.ndarray, params: DispParams | None = None) -> np.ndarray:
    """Oracle strong-pixel mask (uint8 HxW) for a u16/u32/f64 image."""
    image = np.ascontiguousarray(image)
    H, W = image.shape
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    assert mask.shape == (H, W)
    dst = np.empty((H, W), np.uint8)
    fn = {np.dtype(np.uint16): lib().ffs_oracle_dispersion_u16,
          np.dtype(np.uint32): lib().ffs_oracle_dispersion_u32,
          np.dtype(np.float64): lib().ffs_oracle_dispersion_f64}[image.dtype]
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = fn(_ptr(image), _ptr(mask), W, H, C.byref(params) if params else None, _ptr(dst))
    if rc != 0:
        raise MemoryError("oracle dispersion failed")
    return dst


def dispersion_extended(image: np.ndarray, mask: np.ndarray, params: DispParams | None = None, flavour: int = 0,
                        max_valid: float = -1.0, debug: bool = False):
    """Oracle extended-dispersion strong mask (baseline.cpp:730-761).  With debug=True also returns the
    first-pass 'not background' mask and the eroded dispersion mask (1 = signal region)."""
    image = np.ascontiguousarray(image)
    H, W = image.shape
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    assert mask.shape == (H, W)
    dst = np.empty((H, W), np.uint8)
    first = np.empty((H, W), np.uint8) if debug else None
    eroded = np.empty((H, W), np.uint8) if debug else None
    fn = {np.dtype(np.uint16): lib().ffs_oracle_dispersion_extended_u16,
          np.dtype(np.uint32): lib().ffs_oracle_dispersion_extended_u32,
          np.dtype(np.float64): lib().ffs_oracle_dispersion_extended_f64}[image.dtype]
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                   C.c_void_p]
    rc = fn(_ptr(image), _ptr(mask), W, H, C.byref(params) if params else None, flavour, max_valid, _ptr(dst),
            _ptr(first) if debug else None, _ptr(eroded) if debug else None)
    if rc != 0:
        raise MemoryError("oracle extended dispersion failed")
    return (dst, first, eroded) if debug else dst


class PortSpotfinder:
    """Our restatement with the table kept across calls (for timing)."""

    def __init__(self, width: int, height: int):
        self.W, self.H = width, height
        f = lib().ffs_oracle_disp_create
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, C.c_int, C.c_void_p]
        self._h = f(width, height, None)
        lib().ffs_oracle_disp_run.argtypes = [C.c_void_p] * 4
        lib().ffs_oracle_disp_destroy.argtypes = [C.c_void_p]

    def run_f64(self, img_f64: np.ndarray, mask: np.ndarray, dst: np.ndarray) -> None:
        lib().ffs_oracle_disp_run(self._h, _ptr(img_f64), _ptr(mask), _ptr(dst))

    def __del__(self):
        try:
            lib().ffs_oracle_disp_destroy(self._h)
        except Exception:
            pass


class RefSpotfinder:
    """The reference's StandaloneSpotfinder<double> (compiled from its source)."""

    def __init__(self, width: int, height: int):
        self.W, self.H = width, height
        self._h = ref().ffs_ref_create(width, height)

    def __call__(self, image: np.ndarray, mask: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(image, dtype=np.float64)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert img.shape == (self.H, self.W) and mask.shape == (self.H, self.W)
        dst = np.empty((self.H, self.W), np.uint8)
        ref().ffs_ref_standard_dispersion(self._h, _ptr(img), _ptr(mask), self.W, self.H, _ptr(dst))
        return dst

    def run_f64(self, img_f64: np.ndarray, mask: np.ndarray, dst: np.ndarray) -> None:
        """No-conversion entry for timing."""
        ref().ffs_ref_standard_dispersion(self._h, _ptr(img_f64), _ptr(mask), self.W, self.H, _ptr(dst))

    def __del__(self):
        try:
            ref().ffs_ref_destroy(self._h)
        except Exception:
            pass


@dataclass
class CC2D:
    boxes: np.ndarray              # structured: l,t,r,b,num_pixels (after min-size filter)
    n_unfiltered_boxes: int
    num_strong_pixels: int
    num_strong_pixels_filtered: int
    k: np.ndarray                  # uint64 strong linear indices, ascending
    intensity: np.ndarray          # uint32


_BOX_DT = np.dtype([("l", "<u4"), ("t", "<u4"), ("r", "<u4"), ("b", "<u4"), ("num_pixels", "<i4")])
REFL_DT = np.dtype([("x_min", "<u4"), ("x_max", "<u4"), ("y_min", "<u4"), ("y_max", "<u4"),
                    ("z_min", "<i4"), ("z_max", "<i4"), ("num_pixels", "<i4"),
                    ("com_x", "<f4"), ("com_y", "<f4"), ("com_z", "<f4"),
                    ("peak_x", "<u4"), ("peak_y", "<u4"), ("peak_z", "<i4"),
                    ("peak_intensity", "<u4"), ("peak_centroid_distance", "<f4"),
                    ("_pad", "<u4"), ("sum_intensity", "<u8")])
assert REFL_DT.itemsize == C.sizeof(Reflection), (REFL_DT.itemsize, C.sizeof(Reflection))


def cc2d(result_image: np.ndarray, pixels: np.ndarray, min_spot_size: int = 3) -> CC2D:
    result_image = np.ascontiguousarray(result_image, dtype=np.uint8)
    pixels = np.ascontiguousarray(pixels)
    H, W = result_image.shape
    assert pixels.shape == (H, W) and pixels.dtype.itemsize in (2, 4)
    boxes = C.POINTER(Box)()
    nb, nu = C.c_size_t(), C.c_size_t()
    ns, nsf = C.c_uint32(), C.c_uint32()
    ks = C.POINTER(C.c_uint64)()
    it = C.POINTER(C.c_uint32)()
    f = lib().ffs_oracle_cc2d
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(_ptr(result_image), _ptr(pixels), pixels.dtype.itemsize, W, H, min_spot_size,
           C.byref(boxes), C.byref(nb), C.byref(nu), C.byref(ns), C.byref(nsf),
           C.byref(ks), C.byref(it))
    if rc != 0:
        raise MemoryError("oracle cc2d failed")
    n = ns.value
    out_boxes = np.ctypeslib.as_array(C.cast(boxes, C.POINTER(C.c_uint8)), (max(nb.value, 1) * C.sizeof(Box),))[
        : nb.value * C.sizeof(Box)].view(_BOX_DT).copy()
    k = np.ctypeslib.as_array(ks, (max(n, 1),))[:n].copy()
    inten = np.ctypeslib.as_array(it, (max(n, 1),))[:n].copy()
    for p in (boxes, ks, it):
        lib().ffs_oracle_free(C.cast(p, C.c_void_p))
    return CC2D(out_boxes, nu.value, n, nsf.value, k, inten)


@dataclass
class CC3D:
    reflections: np.ndarray        # REFL_DT, label order, after filters
    n_calculated: int
    n_filtered_size: int
    n_filtered_sep: int


def cc3d(slices, width: int, height: int, min_spot_size: int = 3,
         max_peak_centroid_separation: float = 2.0) -> CC3D:
    """slices: list of (k uint64 ascending, intensity uint32) per z."""
    keep = []
    arr = (Slice * max(len(slices), 1))()
    for i, (k, inten) in enumerate(slices):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        inten = np.ascontiguousarray(inten, dtype=np.uint32)
        keep.append((k, inten))
        arr[i].n = len(k)
        arr[i].linear_index = k.ctypes.data_as(C.POINTER(C.c_uint64))
        arr[i].intensity = inten.ctypes.data_as(C.POINTER(C.c_uint32))
    out = C.POINTER(Reflection)()
    n_out, n_calc, n_fs, n_fp = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    f = lib().ffs_oracle_cc3d
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(arr, len(slices), width, height, min_spot_size, max_peak_centroid_separation,
           C.byref(out), C.byref(n_out), C.byref(n_calc), C.byref(n_fs), C.byref(n_fp))
    if rc != 0:
        raise MemoryError("oracle cc3d failed")
    n = n_out.value
    raw = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (max(n, 1) * C.sizeof(Reflection),))
    refl = raw[: n * C.sizeof(Reflection)].view(REFL_DT).copy()
    lib().ffs_oracle_free(C.cast(out, C.c_void_p))
    return CC3D(refl, n_calc.value, n_fs.value, n_fp.value)


class Geometry(C.Structure):
    """ffs_oracle_geometry: mm, pixels, Angstrom, degrees (spotfinder.cc:1157-1166)."""
    _fields_ = [(n, C.c_double) for n in ("distance_mm", "beam_center_x_px", "beam_center_y_px", "pixel_size_x_mm",
                                          "pixel_size_y_mm", "wavelength", "oscillation_start", "oscillation_width")]


def _slices(slices):
    keep = []
    arr = (Slice * max(len(slices), 1))()
    for i, (k, inten) in enumerate(slices):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        inten = np.ascontiguousarray(inten, dtype=np.uint32)
        keep.append((k, inten))
        arr[i].n = len(k)
        arr[i].linear_index = k.ctypes.data_as(C.POINTER(C.c_uint64))
        arr[i].intensity = inten.ctypes.data_as(C.POINTER(C.c_uint32))
    return arr, keep


def cc3d_signals(slices, width, height, min_spot_size=3, max_sep=2.0) -> np.ndarray:
    """Index of the reflection (in cc3d's output) every strong pixel belongs to, vertex order; -1 = filtered."""
    arr, keep = _slices(slices)
    total = sum(len(k) for k, _ in keep)
    out = np.full(max(total, 1), -1, np.int32)
    f = lib().ffs_oracle_cc3d_signals
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    if f(arr, len(slices), width, height, min_spot_size, max_sep, _ptr(out)) != 0:
        raise MemoryError("oracle cc3d failed")
    return out[:total]


def kabsch_variances(slices, width, signal_reflection, reflections, geometry: Geometry):
    """(sigma_b_variance, sigma_m_variance, bbox_depth) per reflection, spotfinder.cc:1152-1215."""
    arr, keep = _slices(slices)
    n = len(reflections)
    sb, sm, depth = np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
    refl = np.zeros(max(n, 1), REFL_DT)
    refl[:n] = reflections
    sig = np.ascontiguousarray(signal_reflection, np.int32)
    f = lib().ffs_oracle_kabsch_variances
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_void_p]
    if f(arr, len(slices), width, _ptr(sig), _ptr(refl), n, C.byref(geometry), _ptr(sb), _ptr(sm), _ptr(depth)) != 0:
        raise MemoryError("oracle kabsch variances failed")
    return sb[:n], sm[:n], depth[:n]


def resolution_mask(mask: np.ndarray, wavelength, distance, beam_center_x, beam_center_y, pixel_size_x, pixel_size_y,
                    dmin=-1.0, dmax=-1.0):
    """(mask after the resolution filter, d-spacing of every pixel as float32): masking.cu:37-147 in float32."""
    H, W = mask.shape
    out = np.ascontiguousarray(mask, np.uint8).copy()
    res = np.empty((H, W), np.float32)
    f = lib().ffs_oracle_resolution_mask
    f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 8 + [C.c_void_p]
    if f(_ptr(out), W, H, wavelength, distance, beam_center_x, beam_center_y, pixel_size_x, pixel_size_y, dmin, dmax,
         _ptr(res)) != 0:
        raise ValueError("oracle resolution mask failed")
    return out, res


def cc2d_reflections(k, intensity, width, height, min_spot_size=3, max_sep=2.0) -> CC3D:
    """find_2d_components (connected_components.cc:238-266): one slice, z = 0."""
    return cc3d([(k, intensity)], width, height, min_spot_size, max_sep)
