"""numpy front end of include/x265_amd_dropin.h: the reference's per-block pixel layer
(inc/x265.h:290-298) with its own names, argument order and HOST buffers, computed on the GPU.

    xSadN[i](N, src, stride_src, ref, stride_ref)          src/pixel.cpp:95-101
    xDctN[i](dst, src, stride, lines, shift)               src/pixel.cpp:571-577
    xInvDctN[i](dst, src, stride, lines, shift)            src/pixel.cpp:579-585
    xSubDct / xIDctAdd / xQuant / xDeQuant                 src/pixel.cpp:591-776

Arrays are flat or 2-D C-contiguous numpy arrays used as raw buffers exactly like the C pointers
(element (y, x) of a block at index y * stride + x).  A failed call raises X265AmdError."""
import numpy as np

from . import api

MODE_INVALID = api.MODE_INVALID


def _ptr(a, dtype, writable=False):
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous):
        raise TypeError(f"expected a C-contiguous numpy array of {np.dtype(dtype)}")
    if writable and not a.flags.writeable:
        raise TypeError("output array is read-only")
    return a.ctypes.data


def _status(where):
    st = api._lib().x265amd_dropin_status()
    if st != 0:
        raise api.X265AmdError(st, where)


def _sad(w):
    def f(N, src, stride_src, ref, stride_ref):
        r = getattr(api._lib(), f"x265amd_xSad{w}xN")(N, _ptr(src, np.uint8), stride_src, _ptr(ref, np.uint8),
                                                      stride_ref)
        _status(f"x265amd_xSad{w}xN")
        return r
    f.__name__ = f"xSad{w}xN"
    return f


def _dct(name):
    def f(dst, src, stride, lines, shift):
        getattr(api._lib(), "x265amd_" + name)(_ptr(dst, np.int16, True), _ptr(src, np.int16), stride, lines, shift)
        _status("x265amd_" + name)
    f.__name__ = name
    return f


xSadN = [_sad(w) for w in (4, 8, 16, 32, 64)]
xDctN = [_dct(n) for n in ("xDST4", "xDCT4", "xDCT8", "xDCT16", "xDCT32")]
xInvDctN = [_dct(n) for n in ("xInvDST4", "xInvDCT4", "xInvDCT8", "xInvDCT16", "xInvDCT32")]


def xSubDct(dst, src, ref, stride, tmp0, tmp1, width, height, mode=MODE_INVALID):
    api._lib().x265amd_xSubDct(_ptr(dst, np.int16, True), _ptr(src, np.uint8), _ptr(ref, np.uint8), stride,
                               None if tmp0 is None else _ptr(tmp0, np.int16),
                               None if tmp1 is None else _ptr(tmp1, np.int16), width, height, mode)
    _status("x265amd_xSubDct")


def xIDctAdd(dst, src, ref, stride, tmp0, tmp1, width, height, mode=MODE_INVALID):
    api._lib().x265amd_xIDctAdd(_ptr(dst, np.uint8, True), _ptr(src, np.int16), _ptr(ref, np.uint8), stride,
                                None if tmp0 is None else _ptr(tmp0, np.int16),
                                None if tmp1 is None else _ptr(tmp1, np.int16), width, height, mode)
    _status("x265amd_xIDctAdd")


def xQuant(dst, src, stride, qp, width, height, slice_type):
    r = api._lib().x265amd_xQuant(_ptr(dst, np.int16, True), _ptr(src, np.int16), stride, qp, width, height,
                                  slice_type)
    _status("x265amd_xQuant")
    return r


def xDeQuant(dst, src, stride, qp, width, height, slice_type=api.SLICE_I):
    api._lib().x265amd_xDeQuant(_ptr(dst, np.int16, True), _ptr(src, np.int16), stride, qp, width, height,
                                slice_type)
    _status("x265amd_xDeQuant")
