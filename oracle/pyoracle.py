"""ctypes doorway to the CPU checker (oracle/liboracle.so) and, when it was
built, to the real reference (oracle/_ref/libx265ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (x265_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODE_INVALID = 255
SLICE_I, SLICE_P = 0, 1

MV_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("sad", "<u4")])

_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)
_u32p = C.POINTER(C.c_uint32)


def build(ref=None):
    """(Re)build liboracle.so, and _ref/libx265ref.so when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref is None:
        ref = os.path.isdir("/root/reference/src")
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _chk(a, dtype):
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a


class _Lib:
    """Common surface of the restatement ("orc_") and the reference shim ("ref_")."""

    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        g = lambda n: getattr(self.lib, prefix + n)
        self._sad = g("sad")
        self._sad.restype = C.c_uint32
        self._sad.argtypes = [C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]
        for n in ("fwd_1d", "inv_1d"):
            f = g(n)
            f.restype = None
            f.argtypes = [C.c_int, _i16p, _i16p, C.c_uint, C.c_int, C.c_int]
        self._sub_dct = g("sub_dct")
        self._sub_dct.restype = None
        self._sub_dct.argtypes = [_i16p, _u8p, _u8p, C.c_uint, _i16p, _i16p, C.c_int, C.c_int, C.c_uint]
        self._idct_add = g("idct_add")
        self._idct_add.restype = None
        self._idct_add.argtypes = [_u8p, _i16p, _u8p, C.c_uint, _i16p, _i16p, C.c_int, C.c_int, C.c_uint]
        self._quant = g("quant")
        self._quant.restype = C.c_uint32
        self._quant.argtypes = [_i16p, _i16p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
        self._dequant = g("dequant")
        self._dequant.restype = None
        self._dequant.argtypes = [_i16p, _i16p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
        self._me = g("me_fullsearch")
        self._me.restype = None
        self._me.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_int, C.c_int, C.c_int, C.c_int]
        self._me_wh = g("me_fullsearch_wh")
        self._me_wh.restype = None
        self._me_wh.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_int, C.c_int, C.c_int, C.c_int]
        self._dct_matrix = g("dct_matrix")
        self._dct_matrix.restype = C.POINTER(C.c_int8)
        self._dct_matrix.argtypes = [C.c_int]
        self._chroma_qp = g("chroma_qp")
        self._chroma_qp.restype = C.c_int
        self._chroma_qp.argtypes = [C.c_int]

    # ---- ROM ----
    def dct_matrix(self, n):
        p = self._dct_matrix(n)
        return np.ctypeslib.as_array(p, shape=(n * n,)).reshape(n, n).copy()

    def chroma_qp(self, qp):
        return self._chroma_qp(qp)

    # ---- per-block primitives (reference signatures, inc/x265.h:290-298) ----
    def sad(self, w, n, src, ss, ref, sr):
        """src/ref: 1-D uint8 buffers (already offset to the block origin)."""
        return self._sad(w, n, _p(_chk(src, np.uint8), _u8p), ss, _p(_chk(ref, np.uint8), _u8p), sr)

    def fwd_1d(self, idx, src, stride, lines, shift):
        dst = np.zeros_like(src)
        getattr(self.lib, self.prefix + "fwd_1d")(idx, _p(dst, _i16p), _p(_chk(src, np.int16), _i16p),
                                                  stride, lines, shift)
        return dst

    def inv_1d(self, idx, src, stride, lines, shift):
        dst = np.zeros_like(src)
        getattr(self.lib, self.prefix + "inv_1d")(idx, _p(dst, _i16p), _p(_chk(src, np.int16), _i16p),
                                                  stride, lines, shift)
        return dst

    def sub_dct(self, src, pred, stride, w, h, mode=MODE_INVALID):
        """Returns coef as int16[h*stride] (in-place tmp0 aliasing like encode.cpp:868)."""
        t0 = np.zeros(h * stride, np.int16)
        t1 = np.zeros(h * stride, np.int16)
        self._sub_dct(_p(t0, _i16p), _p(_chk(src, np.uint8), _u8p), _p(_chk(pred, np.uint8), _u8p),
                      stride, _p(t0, _i16p), _p(t1, _i16p), w, h, mode)
        return t0

    def quant(self, coef, stride, qp, w, h, slice_type):
        dst = np.zeros_like(coef)
        s = self._quant(_p(dst, _i16p), _p(_chk(coef, np.int16), _i16p), stride, qp, w, h, slice_type)
        return dst, s

    def dequant(self, level, stride, qp, w, h, slice_type=SLICE_I):
        dst = np.zeros_like(level)
        self._dequant(_p(dst, _i16p), _p(_chk(level, np.int16), _i16p), stride, qp, w, h, slice_type)
        return dst

    def idct_add(self, coef, pred, stride, w, h, mode=MODE_INVALID):
        rec = np.zeros(h * stride, np.uint8)
        t0 = np.zeros(h * stride, np.int16)
        t1 = np.zeros(h * stride, np.int16)
        src = coef.copy()
        self._idct_add(_p(rec, _u8p), _p(src, _i16p), _p(_chk(pred, np.uint8), _u8p), stride,
                       _p(t0, _i16p), _p(t1, _i16p), w, h, mode)
        return rec

    # ---- frame level ----
    def me_fullsearch(self, cur, ref, ctu, rng, first=0, step=1, count=-1, nthreads=1):
        """cur/ref: uint8[H][W] (stride = W).  Returns MV_DTYPE[nctu] (unvisited CTUs zero)."""
        h, w = cur.shape
        assert ref.shape == cur.shape and w % ctu == 0 and h % ctu == 0
        out = np.zeros((h // ctu) * (w // ctu), MV_DTYPE)
        self._me(_p(_chk(cur, np.uint8), _u8p), _p(_chk(ref, np.uint8), _u8p), w, h, w, ctu, rng,
                 out.ctypes.data, first, step, count, nthreads)
        return out

    def me_fullsearch_wh(self, cur, ref, bw, bh, rng, first=0, step=1, count=-1, nthreads=1):
        """The same for bw x bh blocks tiling the picture.  Returns MV_DTYPE[H / bh][W / bw]."""
        h, w = cur.shape
        assert ref.shape == cur.shape and w % bw == 0 and h % bh == 0
        out = np.zeros(((h // bh), (w // bw)), MV_DTYPE)
        self._me_wh(_p(_chk(cur, np.uint8), _u8p), _p(_chk(ref, np.uint8), _u8p), w, h, w, bw, bh, rng,
                    out.ctypes.data, first, step, count, nthreads)
        return out


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        super().__init__(path, "orc_")
        f = self.lib.orc_tq_forward
        f.restype = None
        f.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, C.c_int, _u8p, _i16p, C.c_int, _u32p, C.c_int]
        f = self.lib.orc_tq_inverse
        f.restype = None
        f.argtypes = [_i16p, C.c_int, _u8p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, C.c_int, _u8p, _u8p, C.c_int, C.c_int]
        f = self.lib.orc_satd
        f.restype = C.c_uint32
        f.argtypes = [C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]

    def satd(self, w, n, src, ss, ref, sr):
        """SATD, HM definition -- not in the reference: PARITY UNPINNED (x265_oracle.c)."""
        return self.lib.orc_satd(w, n, _p(_chk(src, np.uint8), _u8p), ss, _p(_chk(ref, np.uint8), _u8p), sr)

    def tq_forward(self, src, pred, n, qp, slice_type, mv=None, mv_ctu=64, modes=None, nthreads=1):
        h, w = src.shape
        level = np.zeros((h, w), np.int16)
        sums = np.zeros((h // n) * (w // n), np.uint32)
        self.lib.orc_tq_forward(_p(_chk(src, np.uint8), _u8p), w, _p(_chk(pred, np.uint8), _u8p),
                                pred.shape[1], mv.ctypes.data if mv is not None else None, mv_ctu,
                                w, h, n, qp, slice_type, _p(modes, _u8p), _p(level, _i16p), w,
                                _p(sums, _u32p), nthreads)
        return level, sums

    def tq_inverse(self, level, pred, n, qp, slice_type=SLICE_I, mv=None, mv_ctu=64, modes=None,
                   nthreads=1):
        h, w = level.shape
        rec = np.zeros((h, w), np.uint8)
        self.lib.orc_tq_inverse(_p(_chk(level, np.int16), _i16p), w, _p(_chk(pred, np.uint8), _u8p),
                                pred.shape[1], mv.ctypes.data if mv is not None else None, mv_ctu,
                                w, h, n, qp, slice_type, _p(modes, _u8p), _p(rec, _u8p), w, nthreads)
        return rec


    # ---- intra luma (src/encode.cpp:725-765, 1260-1265, 1617-1652) ----
    def intra_filter(self, ref0):
        n = (ref0.size - 1) // 4
        ref1 = np.zeros_like(ref0)
        self.lib.orc_intra_filter(_p(_chk(ref0, np.uint8), _u8p), _p(ref1, _u8p), n)
        return ref1

    def intra_pred_luma(self, ref0, ref1, n, mode):
        dst = np.zeros((n, n), np.uint8)
        self.lib.orc_intra_pred_luma(_p(_chk(ref0, np.uint8), _u8p), _p(_chk(ref1, np.uint8), _u8p), n, mode,
                                     _p(dst, _u8p), n)
        return dst

    def intra_decide_luma(self, src, ref0, mpm, lam):
        """src: uint8[n][n]; ref0: uint8[4n+1]; mpm: uint8[3].  Returns (sads[35], best_mode, best_cost, best_pred)."""
        n = src.shape[0]
        sads = np.zeros(35, np.uint32)
        bm, bc = C.c_uint32(), C.c_uint32()
        pred = np.zeros((n, n), np.uint8)
        self.lib.orc_intra_decide_luma(_p(_chk(src, np.uint8), _u8p), n, _p(_chk(ref0, np.uint8), _u8p),
                                       _p(_chk(mpm, np.uint8), _u8p), C.c_uint32(lam), n,
                                       sads.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(bm), C.byref(bc),
                                       _p(pred, _u8p))
        return sads, bm.value, bc.value, pred


    def intra_encode_frame(self, yuv, w, h, cu, qp, want_levels=False):
        """yuv: uint8[w*h*3/2] (one frame).  Returns (rec uint8[w*h*3/2], mode_y uint8[ncus], mode_c uint8[ncus],
        levels int16[w*h*3/2] or None): the pixel path of the reference's intra encode loop."""
        n = (w // cu) * (h // cu)
        rec = np.zeros_like(yuv)
        my, mc = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        lv = np.zeros(yuv.size, np.int16) if want_levels else None
        self.lib.orc_intra_encode_frame(_p(_chk(yuv, np.uint8), _u8p), w, h, cu, qp, _p(rec, _u8p), _p(my, _u8p),
                                        _p(mc, _u8p), _p(lv, _i16p))
        return rec, my, mc, lv


# oracle/ref_shim.cpp: struct ref_intra_rec (MAX_CU_SIZE = 32, inc/config.h:38)
INTRA_REC_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("ref0", "u1", 129), ("ref1", "u1", 129), ("mpm", "u1", 3),
                            ("best_mode", "u1"), ("sad", "<u4", 35), ("pred_fnv", "<u4", 35)], align=True)


def fnv1a(block):
    """FNV-1a over the bytes of a uint8 block in raster order (the hash ref_shim.cpp records per prediction)."""
    h = 2166136261
    for b in np.ascontiguousarray(block).reshape(-1).tolist():
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h


class Reference(_Lib):
    """The real reference, compiled from /root/reference by `make -C oracle ref`."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libx265ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        super().__init__(path, "ref_")
        f = self.lib.ref_tq_frame
        f.restype = None
        f.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, C.c_int, _u8p, _i16p, C.c_int, _u32p, _u8p, C.c_int]
        f = self.lib.ref_encode_intra
        f.restype = C.c_int
        f.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.POINTER(C.c_int32),
                      _u8p, _u8p, C.c_int]

    def get_tables(self):
        """Current contents of xSadN[], xDctN[], xInvDctN[] as three lists of 5 addresses."""
        arrs = [(C.c_void_p * 5)() for _ in range(3)]
        self.lib.ref_tables_get(*arrs)
        return [list(a) for a in arrs]

    def set_tables(self, sad=None, dct=None, inv=None):
        """Overwrite the reference's mutable plug-in tables (src/pixel.cpp:95,571,579) with function
        addresses; None entries keep what is there."""
        args = [None if t is None else (C.c_void_p * 5)(*t) for t in (sad, dct, inv)]
        self.lib.ref_tables_set(*args)

    def tq_frame(self, src, pred, n, qp, slice_type, mv=None, mv_ctu=64, modes=None):
        h, w = src.shape
        level = np.zeros((h, w), np.int16)
        sums = np.zeros((h // n) * (w // n), np.uint32)
        rec = np.zeros((h, w), np.uint8)
        self.lib.ref_tq_frame(_p(_chk(src, np.uint8), _u8p), w, _p(_chk(pred, np.uint8), _u8p),
                              pred.shape[1], mv.ctypes.data if mv is not None else None, mv_ctu,
                              w, h, n, qp, slice_type, _p(modes, _u8p), _p(level, _i16p), w,
                              _p(sums, _u32p), _p(rec, _u8p), w)
        return level, sums, rec

    def encode_intra(self, yuv, w, h, cu, qp):
        """yuv: uint8[nframes][w*h*3/2].  Returns (rec, bytes_per_frame, md5s, bitstream)."""
        nf = yuv.shape[0]
        rec = np.zeros_like(yuv)
        nbytes = np.zeros(nf, np.int32)
        md5 = np.zeros((nf, 16), np.uint8)
        bs = np.zeros(yuv.size, np.uint8)
        n = self.lib.ref_encode_intra(_p(_chk(yuv, np.uint8), _u8p), w, h, cu, qp, nf, _p(rec, _u8p),
                                      nbytes.ctypes.data_as(C.POINTER(C.c_int32)), _p(md5, _u8p),
                                      _p(bs, _u8p), bs.size)
        if n < 0:
            raise ValueError("reference rejects this geometry (src/params.cpp:73-77)")
        return rec, nbytes, md5, bs[:n].copy()


    def intra_pred_luma(self, ref0, ref1, n, mode):
        """The reference's xEncIntraPredLuma on caller-supplied sample arrays (4n+1 bytes each)."""
        dst = np.zeros((n, n), np.uint8)
        self.lib.ref_intra_pred_luma(_p(_chk(ref0, np.uint8), _u8p), _p(_chk(ref1, np.uint8), _u8p), n, mode,
                                     _p(dst, _u8p))
        return dst

    def intra_record(self, yuv, w, h, cu, qp, frame=0):
        """Run the reference's encoder over frames 0 .. frame and record the luma decisions of `frame` from inside
        its loop.  Returns INTRA_REC_DTYPE[ncus]."""
        cap = (w // cu) * (h // cu)
        out = np.zeros(cap, INTRA_REC_DTYPE)
        assert INTRA_REC_DTYPE.itemsize == 552, INTRA_REC_DTYPE.itemsize
        n = self.lib.ref_intra_record(_p(_chk(yuv, np.uint8), _u8p), w, h, cu, qp, frame,
                                      C.c_void_p(out.ctypes.data), cap)
        if n < 0:
            raise ValueError("reference rejects this geometry")
        return out[:n]


    def write_intra_frames(self, yuv, w, h, cu, qp, rec, levels, mode_y, mode_c):
        """The reference's bitstream writer over a pixel path computed elsewhere: yuv, rec uint8[nf][w*h*3/2],
        levels int16[nf][w*h*3/2], mode_y / mode_c uint8[nf][ncus].  Returns (bytes_per_frame, bitstream)."""
        nf = yuv.shape[0]
        nbytes = np.zeros(nf, np.int32)
        bs = np.zeros(yuv.size + 65536, np.uint8)
        n = self.lib.ref_write_intra_frames(_p(_chk(yuv, np.uint8), _u8p), w, h, cu, qp, nf, _p(_chk(rec, np.uint8), _u8p),
                                            _p(_chk(levels, np.int16), _i16p), _p(_chk(mode_y, np.uint8), _u8p),
                                            _p(_chk(mode_c, np.uint8), _u8p), _p(bs, _u8p), bs.size,
                                            nbytes.ctypes.data_as(C.POINTER(C.c_int32)))
        if n < 0:
            raise ValueError("reference rejects this geometry" if n == -1 else "a chroma mode is not among its CU's candidates")
        return nbytes, bs[:n].copy()


def have_reference():
    return os.path.exists(os.path.join(_HERE, "_ref", "libx265ref.so"))
