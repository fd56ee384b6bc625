"""bitshuffle-LZ4 chunk writer in numpy (+ liblz4 when present) -- makes test / bench input for the GPU
decoder in the detector's wire format (bitshuffle HDF5 filter 32008: 12-byte header, then per block of
8192/elem_size elements a 4-byte big-endian length and an LZ4 block of the bit-transposed elements).
The product never calls this: it is the producer side, which in production is the detector."""
import ctypes
import glob

import numpy as np

_lz4 = None


def liblz4():
    """The system LZ4 (real encoder), or None."""
    global _lz4
    if _lz4 is None:
        _lz4 = False
        for pat in ("/opt/conda/lib/liblz4.so*", "/usr/lib/x86_64-linux-gnu/liblz4.so*", "/usr/lib64/liblz4.so*"):
            for path in sorted(glob.glob(pat)):
                try:
                    lib = ctypes.CDLL(path)
                    lib.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
                    lib.LZ4_compressBound.argtypes = [ctypes.c_int]
                    _lz4 = lib
                    return _lz4
                except OSError:
                    pass
    return _lz4 or None


def lz4_literals_only(block: bytes) -> bytes:
    """A valid LZ4 block with no matches (fallback encoder; also a worst case for the decoder's copy path)."""
    n = len(block)
    out = bytearray()
    if n < 15:
        out.append(n << 4)
    else:
        out.append(0xF0)
        r = n - 15
        while r >= 255:
            out.append(255)
            r -= 255
        out.append(r)
    return bytes(out) + block


def bitshuffle(block: np.ndarray) -> bytes:
    """Elements (multiple of 8) -> es*8 bit planes, element i at byte i/8 bit i%8 of each plane."""
    es = block.dtype.itemsize
    b = block.view(np.uint8).reshape(-1, es)                      # [elem][byte]
    bits = np.unpackbits(b[:, :, None], axis=2, bitorder="little")  # [elem][byte][bit]
    planes = bits.transpose(1, 2, 0).reshape(es * 8, -1)          # [byte*8+bit][elem]
    return np.packbits(planes, axis=1, bitorder="little").tobytes()


def compress(frame: np.ndarray, encoder: str = "auto") -> bytes:
    """One frame -> one chunk.  encoder: "lz4" (system liblz4), "literals", or "auto"."""
    flat = np.ascontiguousarray(frame).reshape(-1)
    es = flat.dtype.itemsize
    n = flat.size
    block = 8192 // es
    lib = liblz4() if encoder in ("auto", "lz4") else None
    if encoder == "lz4" and lib is None:
        raise RuntimeError("no liblz4 on this machine")
    out = bytearray((n * es).to_bytes(8, "big") + (block * es).to_bytes(4, "big"))
    done = 0
    while n - done >= 8:
        this = block if n - done >= block else (n - done) // 8 * 8
        sh = bitshuffle(flat[done:done + this])
        if lib is not None:
            cap = lib.LZ4_compressBound(len(sh))
            dst = ctypes.create_string_buffer(cap)
            clen = lib.LZ4_compress_default(sh, dst, len(sh), cap)
            assert clen > 0
            payload = dst.raw[:clen]
        else:
            payload = lz4_literals_only(sh)
        out += len(payload).to_bytes(4, "big") + payload
        done += this
    out += flat[done:].tobytes()
    return bytes(out)
