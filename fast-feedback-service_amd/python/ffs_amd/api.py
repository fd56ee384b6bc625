"""ctypes binding of include/ffs_hip.h (one class per opaque handle)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lib_path() -> str:
    # FFS_HIP_LIB: an alternative build of the same library (A/B experiments); there is no fallback
    return os.environ.get("FFS_HIP_LIB") or os.path.join(_PKG, "libffs_hip.so")


class FfsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libffs_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """ffs_params (include/ffs_hip.h)."""
    _fields_ = [("min_count", C.c_int32), ("nsig_b", C.c_double), ("nsig_s", C.c_double),
                ("threshold", C.c_double), ("max_valid", C.c_int64),
                ("min_spot_size", C.c_uint32), ("min_spot_size_3d", C.c_uint32),
                ("max_peak_centroid_separation", C.c_float),
                ("want_reflections", C.c_int32), ("want_strong_list", C.c_int32),
                ("want_strong_mask", C.c_int32), ("algorithm", C.c_int32), ("extended_flavour", C.c_int32)]


ALGO_DISPERSION, ALGO_DISPERSION_EXTENDED = 0, 1


class _Box(C.Structure):
    _fields_ = [("l", C.c_uint32), ("t", C.c_uint32), ("r", C.c_uint32), ("b", C.c_uint32),
                ("num_pixels", C.c_int32)]


class _Refl(C.Structure):
    _fields_ = [("x_min", C.c_uint32), ("x_max", C.c_uint32), ("y_min", C.c_uint32), ("y_max", C.c_uint32),
                ("z_min", C.c_int32), ("z_max", C.c_int32), ("num_pixels", C.c_int32),
                ("com_x", C.c_float), ("com_y", C.c_float), ("com_z", C.c_float),
                ("peak_x", C.c_uint32), ("peak_y", C.c_uint32), ("peak_z", C.c_int32),
                ("peak_intensity", C.c_uint32), ("peak_centroid_distance", C.c_float),
                ("flags", C.c_uint32), ("sum_intensity", C.c_uint64)]


class _FrameResult(C.Structure):
    _fields_ = [("frame_id", C.c_int64), ("num_strong_pixels", C.c_uint32),
                ("num_strong_pixels_filtered", C.c_uint32), ("n_components", C.c_uint32),
                ("n_boxes", C.c_uint32), ("boxes", C.POINTER(_Box)),
                ("n_reflections", C.c_uint32), ("reflections", C.POINTER(_Refl)),
                ("n_filtered_size", C.c_uint32), ("n_filtered_sep", C.c_uint32),
                ("strong_k", C.POINTER(C.c_uint32)), ("strong_intensity", C.POINTER(C.c_uint32)),
                ("strong_mask", C.POINTER(C.c_uint8))]


BOX_DT = np.dtype([("l", "<u4"), ("t", "<u4"), ("r", "<u4"), ("b", "<u4"), ("num_pixels", "<i4")])
REFL_DT = np.dtype([("x_min", "<u4"), ("x_max", "<u4"), ("y_min", "<u4"), ("y_max", "<u4"),
                    ("z_min", "<i4"), ("z_max", "<i4"), ("num_pixels", "<i4"),
                    ("com_x", "<f4"), ("com_y", "<f4"), ("com_z", "<f4"),
                    ("peak_x", "<u4"), ("peak_y", "<u4"), ("peak_z", "<i4"),
                    ("peak_intensity", "<u4"), ("peak_centroid_distance", "<f4"),
                    ("flags", "<u4"), ("sum_intensity", "<u8")])
assert REFL_DT.itemsize == C.sizeof(_Refl) and BOX_DT.itemsize == C.sizeof(_Box)
# ffs_frame_result as a numpy record (offsets taken from the ctypes structure: pointers and padding included)
_FRAME_RESULT_DT = np.dtype({"names": ["frame_id", "num_strong_pixels", "num_strong_pixels_filtered", "n_components", "n_boxes", "n_reflections"],
                             "formats": ["<i8", "<u4", "<u4", "<u4", "<u4", "<u4"],
                             "offsets": [getattr(_FrameResult, f).offset for f in
                                         ("frame_id", "num_strong_pixels", "num_strong_pixels_filtered", "n_components", "n_boxes", "n_reflections")],
                             "itemsize": C.sizeof(_FrameResult)})

# every symbol include/ffs_hip.h declares (tests check the library exports them all)
EXPORTS = [
    "ffs_default_params", "ffs_device_count", "ffs_device_name", "ffs_device_total_mem",
    "ffs_ctx_create", "ffs_ctx_destroy", "ffs_last_error", "ffs_ctx_set_mask",
    "ffs_ctx_apply_resolution_mask", "ffs_ctx_get_mask", "ffs_ctx_set_params",
    "ffs_stream_create", "ffs_stream_destroy", "ffs_stream_host_buffer", "ffs_submit",
    "ffs_submit_device", "ffs_ctx_device_layout", "ffs_wait", "ffs_stream_batch_arrays", "ffs_stream_timings",
    "ffs_submit_compressed", "ffs_decode_only", "ffs_stream_spot_centres", "ffs_bench_threshold", "ffs_bench_hbm", "ffs_stream_debug_planes", "ffs_stream_debug_bitplane", "ffs_selftest_sqrt", "ffs_stack3d_create",
    "ffs_stack3d_destroy", "ffs_stack3d_add_batch", "ffs_stack3d_add_slice", "ffs_stack3d_finish", "ffs_stack3d_signals", "ffs_stack3d_last_finish_ms", "ffs_multi_init", "ffs_multi_transport",
    "ffs_ctx_set_tuning", "ffs_bench_pipeline", "ffs_device_numa_node", "ffs_stream_reserve_host", "ffs_stream_last_path", "ffs_multi_gather_rows",
]

_lib = None


def load_library():
    """dlopen libffs_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run `make hip` or __graft_entry__.build(). "
                               "There is no CPU fallback for the product path.")
        L = C.CDLL(p)
        L.ffs_last_error.restype = C.c_char_p
        L.ffs_last_error.argtypes = [C.c_void_p]
        L.ffs_ctx_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_void_p)]
        L.ffs_ctx_destroy.argtypes = [C.c_void_p]
        L.ffs_ctx_set_mask.argtypes = [C.c_void_p, C.c_void_p]
        L.ffs_ctx_get_mask.argtypes = [C.c_void_p, C.c_void_p]
        L.ffs_ctx_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.ffs_ctx_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.ffs_bench_pipeline.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                         C.c_uint32, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ffs_device_numa_node.argtypes = [C.c_int]
        L.ffs_stream_reserve_host.argtypes = [C.c_void_p, C.c_size_t]
        L.ffs_ctx_apply_resolution_mask.argtypes = [C.c_void_p] + [C.c_float] * 8
        L.ffs_ctx_device_layout.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ffs_stream_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.ffs_stream_destroy.argtypes = [C.c_void_p]
        L.ffs_stream_host_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ffs_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64]
        L.ffs_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int64]
        L.ffs_wait.argtypes = [C.c_void_p, C.POINTER(C.POINTER(_FrameResult)), C.POINTER(C.c_uint32)]
        L.ffs_stream_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.ffs_stream_last_path.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ffs_stream_batch_arrays.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.ffs_bench_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                          C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ffs_multi_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_char_p]
        L.ffs_multi_transport.restype = C.c_char_p
        L.ffs_bench_hbm.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ffs_stream_debug_planes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t)]
        L.ffs_stream_spot_centres.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.ffs_submit_compressed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64]
        L.ffs_decode_only.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_float), C.c_void_p]
        L.ffs_stack3d_signals.argtypes = [C.c_void_p] * 7
        L.ffs_stream_debug_bitplane.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.ffs_selftest_sqrt.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.ffs_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.ffs_device_total_mem.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.ffs_stack3d_create.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.ffs_stack3d_destroy.argtypes = [C.c_void_p]
        L.ffs_stack3d_add_batch.argtypes = [C.c_void_p, C.c_void_p]
        L.ffs_stack3d_add_slice.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32]
        L.ffs_stack3d_finish.argtypes = [C.c_void_p, C.POINTER(C.POINTER(_Refl)), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def device_count() -> int:
    return load_library().ffs_device_count()


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    rc = load_library().ffs_device_name(device, buf, 256)
    if rc != 0:
        raise FfsError(rc, "no such device")
    return buf.value.decode()


def default_params() -> Params:
    p = Params()
    load_library().ffs_default_params(C.byref(p))
    return p


@dataclass
class FrameResult:
    frame_id: int
    num_strong_pixels: int
    num_strong_pixels_filtered: int
    n_components: int
    boxes: np.ndarray                      # BOX_DT, label order
    reflections: np.ndarray | None         # REFL_DT, after both filters
    n_filtered_size: int
    n_filtered_sep: int
    strong_k: np.ndarray | None = None
    strong_intensity: np.ndarray | None = None
    strong_mask: np.ndarray | None = None
    extra: dict = field(default_factory=dict)

    @property
    def n_spots_total(self) -> int:        # JSON key, spotfinder.cc:1002
        return len(self.boxes)


def _copy_array(ptr, n, ctype_struct, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dt)
    raw = C.string_at(C.cast(ptr, C.c_void_p), n * C.sizeof(ctype_struct))
    return np.frombuffer(raw, dtype=dt).copy()


class Context:
    def __init__(self, width: int, height: int, dtype=np.uint16, max_batch: int = 1,
                 device: int = 0, max_strong_per_frame: int = 0):
        self._lib = load_library()
        self.W, self.H = int(width), int(height)
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.uint16), np.dtype(np.uint32))
        self.max_batch = int(max_batch)
        self.device = device
        h = C.c_void_p()
        rc = self._lib.ffs_ctx_create(device, self.W, self.H, self.dtype.itemsize, self.max_batch,
                                      max_strong_per_frame, C.byref(h))
        if rc != 0:
            raise FfsError(rc, self._lib.ffs_last_error(None).decode())
        self._h = h
        self.params = default_params()

    def _check(self, rc):
        if rc != 0:
            raise FfsError(rc, self._lib.ffs_last_error(self._h).decode())

    def set_mask(self, mask: np.ndarray | None):
        if mask is None:
            self._check(self._lib.ffs_ctx_set_mask(self._h, None))
            return
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.H, self.W)
        self._check(self._lib.ffs_ctx_set_mask(self._h, m.ctypes.data_as(C.c_void_p)))

    def get_mask(self) -> np.ndarray:
        m = np.empty((self.H, self.W), np.uint8)
        self._check(self._lib.ffs_ctx_get_mask(self._h, m.ctypes.data_as(C.c_void_p)))
        return m

    def apply_resolution_mask(self, wavelength, distance_m, beam_center_x_px, beam_center_y_px,
                              pixel_size_x_m, pixel_size_y_m, dmin=-1.0, dmax=-1.0):
        self._check(self._lib.ffs_ctx_apply_resolution_mask(
            self._h, wavelength, distance_m, beam_center_x_px, beam_center_y_px,
            pixel_size_x_m, pixel_size_y_m, dmin, dmax))

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        self._check(self._lib.ffs_ctx_set_params(self._h, C.byref(self.params)))

    def set_tuning(self, **kw):
        """ffs_ctx_set_tuning: A/B partners, fall-backs and capacities (same results either way); see include/ffs_hip.h."""
        for k, v in kw.items():
            self._check(self._lib.ffs_ctx_set_tuning(self._h, k.encode(), int(v)))

    def selftest_sqrt(self, begin: int, end: int) -> int:
        out = C.c_uint64()
        self._check(self._lib.ffs_selftest_sqrt(self._h, begin, end, C.byref(out)))
        return out.value

    def device_layout(self):
        pitch, stride = C.c_size_t(), C.c_size_t()
        self._check(self._lib.ffs_ctx_device_layout(self._h, C.byref(pitch), C.byref(stride)))
        return pitch.value, stride.value

    def stream(self) -> "Stream":
        return Stream(self)

    def close(self):
        if self._h:
            self._lib.ffs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        ctx._check(self._lib.ffs_stream_create(ctx._h, C.byref(h)))
        self._h = h

    def host_buffer(self) -> np.ndarray:
        """The stream's pinned staging area as a (max_batch, H, W) array."""
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx._check(self._lib.ffs_stream_host_buffer(self._h, C.byref(p), C.byref(n)))
        frames = self.ctx.max_batch * self.ctx.H * self.ctx.W * self.ctx.dtype.itemsize  # (+ slack for chunks)
        assert n.value >= frames, "the staging area was reserved smaller than max_batch frames (reserve_host)"
        buf = (C.c_uint8 * frames).from_address(p.value)
        return np.frombuffer(buf, dtype=self.ctx.dtype).reshape(self.ctx.max_batch, self.ctx.H, self.ctx.W)

    def reserve_host(self, nbytes: int):
        """Size the pinned staging area before its first use (ffs_stream_reserve_host), e.g. for chunks instead of frames."""
        self.ctx._check(self._lib.ffs_stream_reserve_host(self._h, nbytes))

    def host_bytes(self) -> np.ndarray:
        """The same staging area as bytes, including its slack (for compressed chunks placed in it)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx._check(self._lib.ffs_stream_host_buffer(self._h, C.byref(p), C.byref(n)))
        return np.frombuffer((C.c_uint8 * n.value).from_address(p.value), dtype=np.uint8)

    def _chunk_args(self, chunks):
        keep = [np.frombuffer(c, np.uint8) if not isinstance(c, np.ndarray) else c for c in chunks]
        ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
        sizes = (C.c_size_t * len(keep))(*[k.size for k in keep])
        return keep, ptrs, sizes

    def submit_compressed(self, chunks, first_frame_id: int = 0):
        """chunks: bitshuffle-LZ4 chunks (bytes / uint8 arrays, 12-byte header included), one per frame."""
        self._keep, ptrs, sizes = self._chunk_args(chunks)
        self.ctx._check(self._lib.ffs_submit_compressed(self._h, ptrs, sizes, len(self._keep), first_frame_id))

    def process_compressed(self, chunks, first_frame_id: int = 0):
        self.submit_compressed(chunks, first_frame_id)
        return self.wait()

    def decode_only(self, chunks, iters: int = 1, want_frames: bool = True):
        """GPU decode alone: (average ms per launch, decoded frames or None)."""
        keep, ptrs, sizes = self._chunk_args(chunks)
        ms = C.c_float()
        out = np.empty((len(keep), self.ctx.H, self.ctx.W), self.ctx.dtype) if want_frames else None
        self.ctx._check(self._lib.ffs_decode_only(self._h, ptrs, sizes, len(keep), iters, C.byref(ms),
                                                  out.ctypes.data_as(C.c_void_p) if want_frames else None))
        return ms.value, out

    def submit(self, frames: np.ndarray, first_frame_id: int = 0):
        f = np.ascontiguousarray(frames, dtype=self.ctx.dtype)
        if f.ndim == 2:
            f = f[None]
        assert f.shape[1:] == (self.ctx.H, self.ctx.W), f.shape
        self._keep = f
        self.ctx._check(self._lib.ffs_submit(self._h, f.ctypes.data_as(C.c_void_p), f.shape[0], first_frame_id))

    def submit_device(self, dev_ptr: int, pitch_bytes: int, frame_stride_bytes: int, n_frames: int,
                      first_frame_id: int = 0):
        self.ctx._check(self._lib.ffs_submit_device(self._h, C.c_void_p(dev_ptr), pitch_bytes,
                                                    frame_stride_bytes, n_frames, first_frame_id))

    def wait(self, copy: bool = True) -> list[FrameResult]:
        """copy=False: the box / reflection arrays are views of the library's buffers, valid until the
        next wait() on this stream (what a C caller gets); copy=True detaches them."""
        res = C.POINTER(_FrameResult)()
        n = C.c_uint32()
        self.ctx._check(self._lib.ffs_wait(self._h, C.byref(res), C.byref(n)))
        # two copies for the whole batch, then per-frame views
        bp, rp = C.c_void_p(), C.c_void_p()
        nb, nr = C.c_uint32(), C.c_uint32()
        self.ctx._check(self._lib.ffs_stream_batch_arrays(self._h, C.byref(bp), C.byref(nb),
                                                          C.byref(rp), C.byref(nr)))
        def arr(ptr, count, dt):
            if not count:
                return np.zeros(0, dt)
            buf = (C.c_uint8 * (count * dt.itemsize)).from_address(ptr.value)
            a = np.frombuffer(buf, dt)
            return a.copy() if copy else a
        boxes, refls = arr(bp, nb.value, BOX_DT), arr(rp, nr.value, REFL_DT)
        self.last_batch_boxes, self.last_batch_reflections = boxes, refls   # whole-batch arrays
        want_refl = bool(self.ctx.params.want_reflections)
        want_list = bool(self.ctx.params.want_strong_list)
        out = []
        W, H = self.ctx.W, self.ctx.H
        b0 = r0 = 0
        for i in range(n.value):
            r = res[i]
            ns = r.num_strong_pixels
            fr = FrameResult(
                frame_id=r.frame_id, num_strong_pixels=ns,
                num_strong_pixels_filtered=r.num_strong_pixels_filtered,
                n_components=r.n_components,
                boxes=boxes[b0:b0 + r.n_boxes],
                reflections=refls[r0:r0 + r.n_reflections] if want_refl else None,
                n_filtered_size=r.n_filtered_size, n_filtered_sep=r.n_filtered_sep)
            b0 += r.n_boxes
            r0 += r.n_reflections
            if want_list:
                if ns and r.strong_k:
                    fr.strong_k = np.ctypeslib.as_array(r.strong_k, (ns,)).copy()
                    fr.strong_intensity = np.ctypeslib.as_array(r.strong_intensity, (ns,)).copy()
                else:
                    fr.strong_k = np.zeros(0, np.uint32)
                    fr.strong_intensity = np.zeros(0, np.uint32)
            if r.strong_mask:
                fr.strong_mask = np.ctypeslib.as_array(r.strong_mask, (H, W)).copy()
            out.append(fr)
        return out

    def wait_counts(self):
        """ffs_wait() for callers that want the batch's totals only: (frames, boxes, strong pixels), read from the library's
        result array in one vectorised pass -- no Python object per frame (32 FrameResult objects cost the caller ~100 us,
        which the last wait of a timed run pays on the clock).  The boxes and reflections are in the library's host arrays
        (ffs_stream_batch_arrays) until the next wait on this stream."""
        res = C.POINTER(_FrameResult)()
        n = C.c_uint32()
        self.ctx._check(self._lib.ffs_wait(self._h, C.byref(res), C.byref(n)))
        if not n.value:
            return 0, 0, 0
        raw = np.frombuffer((C.c_uint8 * (n.value * C.sizeof(_FrameResult))).from_address(C.addressof(res.contents)), _FRAME_RESULT_DT)
        self.last_frame_counts = raw     # per-frame view (n_boxes, num_strong_pixels, ...), valid until the next wait on this stream
        return int(n.value), int(raw["n_boxes"].sum()), int(raw["num_strong_pixels"].sum())

    def process(self, frames: np.ndarray, first_frame_id: int = 0) -> list[FrameResult]:
        self.submit(frames, first_frame_id)
        return self.wait()

    def timings(self):
        t = (C.c_float * 5)()
        self.ctx._check(self._lib.ffs_stream_timings(self._h, t))
        return dict(zip(("h2d", "threshold", "ccl", "d2h", "total"), list(t)))

    PATH_BITS = {"wave_logs": 1, "frame_chain": 2, "bands": 4, "runs": 8, "grid_kernels": 16, "extended": 32}

    def last_path(self):
        """ffs_stream_last_path: (set of the launches the last batch took, times ffs_wait ran it again)."""
        bits, reruns = C.c_uint32(), C.c_uint32()
        self.ctx._check(self._lib.ffs_stream_last_path(self._h, C.byref(bits), C.byref(reruns)))
        return {k for k, v in self.PATH_BITS.items() if bits.value & v}, reruns.value

    def bench_threshold(self, dev_ptr: int, pitch_bytes: int, frame_stride_bytes: int, n_frames: int,
                        iters: int):
        a, b = C.c_float(), C.c_float()
        self.ctx._check(self._lib.ffs_bench_threshold(self._h, C.c_void_p(dev_ptr), pitch_bytes,
                                                      frame_stride_bytes, n_frames, iters,
                                                      C.byref(a), C.byref(b)))
        return a.value, b.value

    def bench_hbm(self, iters: int = 5):
        """(read-only GB/s, 2:1 read/write GB/s) measured on this stream's buffers (ffs_bench_hbm)."""
        a, b = C.c_float(), C.c_float()
        self.ctx._check(self._lib.ffs_bench_hbm(self._h, iters, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pack_spot_centres(self, out: np.ndarray, cap: int) -> int:
        """Rows (frame_id, x, y, z) of the last batch's reflections into `out` ((cap + 1, 4) float32,
        last row = count); the layout dist.pack_spots produces."""
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= (cap + 1) * 4
        n = C.c_uint32()
        self.ctx._check(self._lib.ffs_stream_spot_centres(self._h, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return n.value

    def debug_bitplane(self, frame_in_batch: int, which: int) -> np.ndarray:
        """0 = strong plane, 1 = extended first pass, 2 = extended eroded signal region (H x W uint8)."""
        out = np.empty((self.ctx.H, self.ctx.W), np.uint8)
        self.ctx._check(self._lib.ffs_stream_debug_bitplane(self._h, frame_in_batch, which,
                                                            out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._h:
            self._lib.ffs_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bench_pipeline(streams, dev_ptr: int, pitch_bytes: int, frame_stride_bytes: int, n_frames: int, steps: int,
                   first_frame_id: int = 0):
    """ffs_bench_pipeline: `steps` batches through `streams` (one context), natively; -> (boxes, strong pixels) summed."""
    arr = (C.c_void_p * len(streams))(*[s._h for s in streams])
    nb, ns = C.c_uint64(), C.c_uint64()
    streams[0].ctx._check(load_library().ffs_bench_pipeline(arr, len(streams), C.c_void_p(dev_ptr), pitch_bytes, frame_stride_bytes,
                                                             n_frames, steps, first_frame_id, C.byref(nb), C.byref(ns)))
    return nb.value, ns.value


def device_numa_node(device: int = 0) -> int:
    return load_library().ffs_device_numa_node(device)


def multi_init(devices, transport=None) -> str:
    """ffs_multi_init: prepare the exchange between the contexts of several GPUs; returns the transport in use."""
    lib = load_library()
    arr = (C.c_int * len(devices))(*devices)
    rc = lib.ffs_multi_init(arr, len(devices), transport.encode() if transport else None)
    if rc != 0:
        raise FfsError(rc, lib.ffs_last_error(None).decode())
    return lib.ffs_multi_transport().decode()


def multi_gather_rows(streams, root: int = 0, cap: int = 1 << 20) -> np.ndarray:
    """ffs_multi_gather_rows: the centre rows of the streams' last batches through RCCL (counts by all-gather, rows by send / recv to
    the root's device, one copy to the host) -> (n, 4) float32, lane 0 = frame id bits, in the order of `streams`."""
    lib = load_library()
    lib.ffs_multi_gather_rows.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    arr = (C.c_void_p * len(streams))(*[s._h for s in streams])
    out = np.empty((cap, 4), np.float32)
    n = C.c_uint32()
    streams[root].ctx._check(lib.ffs_multi_gather_rows(arr, len(streams), root, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
    return out[:n.value].copy()


class Stack3D:
    """Rotation-sweep accumulator -> 3D connected components (ffs_stack3d_*)."""

    def __init__(self, ctx: Context, max_total_strong: int = 0):
        self.ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        ctx._check(self._lib.ffs_stack3d_create(ctx._h, max_total_strong, C.byref(h)))
        self._h = h

    def add_batch(self, stream: Stream):
        self.ctx._check(self._lib.ffs_stack3d_add_batch(self._h, stream._h))

    def add_slice(self, frame_id: int, k: np.ndarray, intensity: np.ndarray):
        k = np.ascontiguousarray(k, dtype=np.uint32)
        it = np.ascontiguousarray(intensity, dtype=np.uint32)
        assert k.shape == it.shape
        self.ctx._check(self._lib.ffs_stack3d_add_slice(self._h, frame_id, k.ctypes.data_as(C.c_void_p),
                                                        it.ctypes.data_as(C.c_void_p), len(k)))

    def finish(self):
        """-> (reflections REFL_DT, n_calculated, n_filtered_size, n_filtered_sep)"""
        r = C.POINTER(_Refl)()
        n, nc, fs, fp = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.ctx._check(self._lib.ffs_stack3d_finish(self._h, C.byref(r), C.byref(n), C.byref(nc),
                                                     C.byref(fs), C.byref(fp)))
        return _copy_array(r, n.value, _Refl, REFL_DT), nc.value, fs.value, fp.value

    def last_finish_ms(self) -> float:
        ms = C.c_float()
        self.ctx._check(self._lib.ffs_stack3d_last_finish_ms(self._h, C.byref(ms)))
        return ms.value

    def signals(self):
        """Per-signal view of the last finish(): dict of x, y, z, intensity, reflection (-1 = filtered)."""
        px, py, pi = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        pz, pr = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        n = C.c_uint64()
        self.ctx._check(self._lib.ffs_stack3d_signals(self._h, C.byref(px), C.byref(py), C.byref(pz), C.byref(pi),
                                                      C.byref(pr), C.byref(n)))
        def arr(p, dt):
            return np.ctypeslib.as_array(p, (n.value,)).astype(dt) if n.value else np.zeros(0, dt)
        return {"x": arr(px, np.uint32), "y": arr(py, np.uint32), "z": arr(pz, np.int32),
                "intensity": arr(pi, np.uint32), "reflection": arr(pr, np.int32)}

    def close(self):
        if self._h:
            self._lib.ffs_stack3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
