"""ctypes binding of include/x265_amd.h plus a thin host layer that mirrors the
reference's pixel-layer interface (inc/x265.h:290-298) at picture granularity.

PyTorch is used for device memory and streams only.  There is no CPU path here:
if lib/libx265amd.so is missing or the device is not gfx950, calls raise.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("X265AMD_LIB_PATH") or os.path.join(_HERE, "lib", "libx265amd.so")  # the override: A/B runs of tools/

MODE_INVALID = 255            # inc/x265.h:70
SLICE_I, SLICE_P, SLICE_B = 0, 1, 2   # inc/x265.h:39-43
TQ_CHECK_MV = 1                        # x265amd_tq_desc.flags: verify the vectors before transforming
MAX_SAD = 1 << 30             # inc/config.h:52

MV_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("sad", "<u4")])


class X265AmdError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib().x265amd_strerror(status).decode() if _LIB is not None else "library not loaded"
        super().__init__(f"{where}: {msg} (status {status})")


class TqDesc(C.Structure):
    """x265amd_tq_desc, include/x265_amd.h"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("tu", C.c_int32), ("qp", C.c_int32),
        ("slice_type", C.c_int32), ("nframes", C.c_int32),
        ("d_src", C.c_void_p), ("src_stride", C.c_int32), ("src_frame_stride", C.c_int64),
        ("d_pred", C.c_void_p), ("pred_stride", C.c_int32), ("pred_frame_stride", C.c_int64),
        ("d_mv", C.c_void_p), ("mv_ctu", C.c_int32), ("mv_frame_stride", C.c_int64),
        ("d_modes", C.c_void_p), ("modes_frame_stride", C.c_int64),
        ("d_level", C.c_void_p), ("level_stride", C.c_int32), ("level_frame_stride", C.c_int64),
        ("d_coef", C.c_void_p),
        ("d_abs_sum", C.c_void_p), ("abs_sum_frame_stride", C.c_int64),
        ("d_rec", C.c_void_p), ("rec_stride", C.c_int32), ("rec_frame_stride", C.c_int64),
        ("mv_shift", C.c_int32), ("flags", C.c_int32),
    ]


_LIB = None

# name -> (restype, argtypes); every symbol include/x265_amd.h declares
SIGNATURES = {
    "x265amd_init": (C.c_int, [C.c_int]),
    "x265amd_version": (C.c_int, []),
    "x265amd_strerror": (C.c_char_p, [C.c_int]),
    "x265amd_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "x265amd_free": (C.c_int, [C.c_void_p]),
    "x265amd_malloc_host": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "x265amd_free_host": (C.c_int, [C.c_void_p]),
    "x265amd_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "x265amd_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "x265amd_stream_sync": (C.c_int, [C.c_void_p]),
    "x265amd_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "x265amd_stream_destroy": (C.c_int, [C.c_void_p]),
    "x265amd_chroma_qp": (C.c_int, [C.c_int]),
    "x265amd_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "x265amd_pack_levels_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "x265amd_pack_levels": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "x265amd_sad_blocks": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "x265amd_satd_blocks": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "x265amd_me_fullsearch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                         C.c_void_p]),
    "x265amd_me_fullsearch_pu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "x265amd_me_fullsearch_pu_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_void_p),
                                               C.c_void_p]),
    "x265amd_intra_decide_luma": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "x265amd_intra_waves": (C.c_int, [C.c_void_p]),
    "x265amd_intra_predict_wave": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "x265amd_intra_encode_frame": (C.c_int, [C.c_void_p, C.c_void_p]),
    "x265amd_intra_encode_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "x265amd_intra_stream_bound": (C.c_size_t, [C.c_int, C.c_int]),
    "x265amd_intra_write_picture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "x265amd_me_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "x265amd_me_workspace_bytes_ctu": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "x265amd_me_fullsearch_sea": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]),
    "x265amd_me_static_workspace_bytes": (C.c_size_t, []),
    "x265amd_me_fullsearch_sea_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                                    C.c_void_p]),
    "x265amd_tq_forward": (C.c_int, [C.POINTER(TqDesc), C.c_void_p]),
    "x265amd_tq_inverse": (C.c_int, [C.POINTER(TqDesc), C.c_void_p]),
    "x265amd_idct_add": (C.c_int, [C.POINTER(TqDesc), C.c_void_p]),
    "x265amd_tq_roundtrip": (C.c_int, [C.POINTER(TqDesc), C.c_void_p]),
    "x265amd_quant_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "x265amd_dequant_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
}

# include/x265_amd_dropin.h: the reference's own per-block signatures (inc/x265.h:290-298), host pointers
_SAD = (C.c_uint32, [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint])
_DCT = (None, [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int])
_BLK = (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint])
# include/x265_amd_debug.h: instrumentation hooks (tools/, bench.py, tests), not part of the drop-in ABI
DEBUG_SIGNATURES = {
    "x265amd_debug_set_me_stats": (None, [C.c_void_p]),
    "x265amd_debug_me_timing": (C.c_int, [C.c_int]),
    "x265amd_debug_me_last_ms": (C.c_int, [C.POINTER(C.c_float)]),
    "x265amd_debug_sea_shape": (None, [C.c_int, C.c_int]),
    "x265amd_debug_tq_path": (None, [C.c_int]),
    "x265amd_debug_intra_fused": (None, [C.c_int]),
    "x265amd_debug_intra_threads": (None, [C.c_int]),
    "x265amd_debug_set_intra_stats": (None, [C.c_void_p]),
    "x265amd_debug_tq_trace": (None, [C.c_void_p, C.c_int]),
    "x265amd_debug_tq_units_per_wg": (None, [C.c_int]),
    "x265amd_debug_me_fullsearch_basic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                                     C.c_void_p]),
}

DROPIN_SIGNATURES = {
    "x265amd_dropin_status": (C.c_int, []),
    "x265amd_dropin_calls": (C.c_uint64, []),
    "x265amd_dropin_teardown": (C.c_int, []),
    **{f"x265amd_xSad{w}xN": _SAD for w in (4, 8, 16, 32, 64)},
    **{f"x265amd_{n}": _DCT for n in ("xDST4", "xDCT4", "xDCT8", "xDCT16", "xDCT32",
                                       "xInvDST4", "xInvDCT4", "xInvDCT8", "xInvDCT16", "xInvDCT32")},
    "x265amd_xSubDct": _BLK,
    "x265amd_xIDctAdd": _BLK,
    "x265amd_xQuant": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]),
    "x265amd_xDeQuant": (None, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]),
}


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C x265_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **DROPIN_SIGNATURES, **DEBUG_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def load_library():
    """Load lib/libx265amd.so and bind every declared symbol (no GPU needed)."""
    return _lib()


def _check(status, where):
    if status != 0:
        raise X265AmdError(status, where)


def init(device=0):
    """x265amd_init: bind this process to `device` (must be gfx950)."""
    _check(_lib().x265amd_init(int(device)), "x265amd_init")
    torch.cuda.set_device(device)


def chroma_qp(qp):
    r = _lib().x265amd_chroma_qp(int(qp))
    if r < 0:
        raise X265AmdError(r, "x265amd_chroma_qp")
    return r


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype):
        raise TypeError(f"{name}: expected a CUDA tensor of {dtype}, got {type(t)} {getattr(t, 'dtype', None)}")
    return t


def _planes(t, dtype, name):
    """[H][W] or [F][H][W] tensor with unit pixel stride -> (ptr, nframes, H, W, row stride, frame stride)."""
    _dev(t, dtype, name)
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError(f"{name}: need [F][H][W] with contiguous rows")
    f, h, w = t.shape
    return t.data_ptr(), f, h, w, t.stride(1), (t.stride(0) if f > 1 else 0)


# --------------------------------------------------------------------------
# xSadN[] (inc/x265.h:290-291)
# --------------------------------------------------------------------------
def sad_blocks(w, n, src, stride_src, ref, stride_ref, src_off, ref_off):
    """Batched xSadN[log2(w)-2](n, src+src_off[b], stride_src, ref+ref_off[b], stride_ref)."""
    _dev(src, torch.uint8, "src"), _dev(ref, torch.uint8, "ref")
    _dev(src_off, torch.int32, "src_off"), _dev(ref_off, torch.int32, "ref_off")
    nb = src_off.numel()
    if ref_off.numel() != nb:
        raise ValueError("src_off / ref_off length mismatch")
    out = torch.empty(nb, dtype=torch.int32, device=src.device)
    _check(_lib().x265amd_sad_blocks(w, n, src.data_ptr(), stride_src, ref.data_ptr(), stride_ref,
                                     src_off.data_ptr(), ref_off.data_ptr(), nb, out.data_ptr(), _stream()),
           "x265amd_sad_blocks")
    return out


def satd_blocks(w, n, src, stride_src, ref, stride_ref, src_off, ref_off):
    """Batched SATD (HM definition; not in the reference, parity unpinned) with sad_blocks' convention."""
    _dev(src, torch.uint8, "src"), _dev(ref, torch.uint8, "ref")
    _dev(src_off, torch.int32, "src_off"), _dev(ref_off, torch.int32, "ref_off")
    nb = src_off.numel()
    if ref_off.numel() != nb:
        raise ValueError("src_off / ref_off length mismatch")
    out = torch.empty(nb, dtype=torch.int32, device=src.device)
    _check(_lib().x265amd_satd_blocks(w, n, src.data_ptr(), stride_src, ref.data_ptr(), stride_ref,
                                      src_off.data_ptr(), ref_off.data_ptr(), nb, out.data_ptr(), _stream()),
           "x265amd_satd_blocks")
    return out


def pack_levels(level, tu, abs_sum=None, capacity=None, out=None, begin=True):
    """x265amd_pack_levels on a level plane [H][W] or batch [F][H][W] (int16, on the GPU): per TU where its non-zero
    levels are and how many, and the levels as (position << 16 | level) words.  Returns (table int32 [F][ntu][2],
    words int32 [capacity], cursor int32 [1]); `out` = those three to reuse, `begin=False` to go on filling them."""
    lp, f, h, w, stride, fs = _planes(level, torch.int16, "level")
    ntu = (h // tu) * (w // tu) if tu > 0 else 0
    if out is None:
        cap = f * h * w if capacity is None else int(capacity)
        out = (torch.empty((f, ntu, 2), dtype=torch.int32, device=level.device),
               torch.empty(max(cap, 1), dtype=torch.int32, device=level.device),
               torch.zeros(1, dtype=torch.int32, device=level.device))
    table, words, cursor = out
    cap = words.numel() if capacity is None else int(capacity)
    if table.numel() < f * ntu * 2 or cap > words.numel():
        raise ValueError("pack_levels: table / words too small")
    sp, sfs = 0, 0
    if abs_sum is not None:
        _dev(abs_sum, torch.int32, "abs_sum")
        if not abs_sum.is_contiguous() or abs_sum.numel() < f * ntu:
            raise ValueError("abs_sum: need a contiguous int32 tensor of [F][ntu]")
        sp, sfs = abs_sum.data_ptr(), ntu
    if begin:
        _check(_lib().x265amd_pack_levels_begin(cursor.data_ptr(), _stream()), "x265amd_pack_levels_begin")
    _check(_lib().x265amd_pack_levels(lp, w, h, stride, tu, f, fs, sp, sfs, table.data_ptr(), words.data_ptr(), cap,
                                      cursor.data_ptr(), _stream()), "x265amd_pack_levels")
    return table, words, cursor


def unpack_levels(table, words, width, height, tu):
    """The dense planes back from pack_levels' output (host side, numpy): table [F][ntu][2], words [n]."""
    table = np.asarray(table).astype(np.int64)
    words = np.ascontiguousarray(words)
    words = words.view(np.uint32) if words.dtype == np.int32 else words.astype(np.uint32)
    f, ntu = table.shape[0], table.shape[1]
    tus_x = width // tu
    out = np.zeros((f, height, width), np.int16)
    for fi in range(f):
        for t in range(ntu):
            off, cnt = table[fi, t]
            if cnt == 0:
                continue
            wd = words[off:off + cnt]
            pos = (wd >> 16).astype(np.int64)
            ty, tx = divmod(t, tus_x)
            out[fi, ty * tu + pos // tu, tx * tu + pos % tu] = (wd & 0xffff).astype(np.uint16).view(np.int16)
    return out


# --------------------------------------------------------------------------
# motion search
# --------------------------------------------------------------------------
def me_workspace_bytes(width, height, search_range, nframes, nrefs, ctu=64):
    if ctu != 64:
        return int(_lib().x265amd_me_workspace_bytes_ctu(width, height, ctu, search_range, nframes, nrefs))
    return int(_lib().x265amd_me_workspace_bytes(width, height, search_range, nframes, nrefs))


def me_static_workspace_bytes():
    return int(_lib().x265amd_me_static_workspace_bytes())


def me_fullsearch(cur, ref, ctu=64, search_range=64, out=None, method="auto", workspace=None):
    """cur: uint8 [F][H][W]; ref: uint8 [F][H][W] (one reference per frame) or
    [F][R][H][W] (R references per frame).  Returns int32 [F][R][nctu][2] viewed as
    (mvx|mvy<<16, sad) -- use mv_to_numpy() for a structured host copy.

    method: "brute" = x265amd_me_fullsearch (every candidate's SAD), "sea" =
    x265amd_me_fullsearch_sea (exact successive elimination, same result; CTU 64 or 32;
    `workspace` = uint8 CUDA tensor of me_workspace_bytes() or None to allocate one),
    "sea_static" = x265amd_me_fullsearch_sea_static (the same search on the library's static scratch: the
    call allocates nothing, `out` must be given; jobs beyond me_static_workspace_bytes() are refused),
    "auto" = "sea" where it applies; "basic" = the independent one-candidate-per-lane kernel (tests)."""
    cp, f, h, w, cs, cfs = _planes(cur, torch.uint8, "cur")
    _dev(ref, torch.uint8, "ref")
    if ref.dim() == 2:
        ref = ref.unsqueeze(0)
    if ref.dim() == 3:
        ref = ref.unsqueeze(1)
    if ref.dim() != 4 or ref.shape[0] != f or ref.shape[2:] != (h, w) or ref.stride(3) != 1 or ref.stride(2) != cs:
        raise ValueError("ref: need [F][R][H][W] with the row stride of cur")
    nrefs = ref.shape[1]
    nctu = (h // ctu) * (w // ctu) if ctu > 0 else 0
    if out is None:
        out = torch.empty((f, nrefs, max(nctu, 0), 2), dtype=torch.int32, device=cur.device)
    _dev(out, torch.int32, "out")
    if not out.is_contiguous() or out.numel() < f * nrefs * max(nctu, 0) * 2:
        raise ValueError("out: need a contiguous int32 tensor of at least [F][R][nctu][2] elements")
    if method not in ("auto", "brute", "sea", "sea_static", "basic"):
        raise ValueError("method")
    if method == "sea_static":
        _check(_lib().x265amd_me_fullsearch_sea_static(cp, ref.data_ptr(), w, h, cs, ctu, search_range, f, cfs,
                                                       ref.stride(0) if f > 1 else 0, nrefs,
                                                       ref.stride(1) if nrefs > 1 else 0, out.data_ptr(), _stream()),
               "x265amd_me_fullsearch_sea_static")
        return out
    if method == "basic":  # the one-candidate-per-lane cross-check kernel (debug entry point, not in the header)
        fn = _lib().x265amd_debug_me_fullsearch_basic
        _check(fn(cp, ref.data_ptr(), w, h, cs, ctu, search_range, f, cfs, ref.stride(0) if f > 1 else 0, nrefs,
                  ref.stride(1) if nrefs > 1 else 0, out.data_ptr(), _stream()), "x265amd_debug_me_fullsearch_basic")
        return out
    use_sea = method == "sea" or (method == "auto" and ctu == 64 and w > 0 and h > 0 and w % 64 == 0 and h % 64 == 0)
    if use_sea:
        need = me_workspace_bytes(w, h, search_range, f, nrefs, ctu)
        if workspace is None:
            workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=cur.device)
        _dev(workspace, torch.uint8, "workspace")
        st = _lib().x265amd_me_fullsearch_sea(cp, ref.data_ptr(), w, h, cs, ctu, search_range, f, cfs,
                                              ref.stride(0) if f > 1 else 0, nrefs,
                                              ref.stride(1) if nrefs > 1 else 0, out.data_ptr(),
                                              workspace.data_ptr(), workspace.numel(), _stream())
        if st == -5 and method == "auto":   # X265AMD_ERR_LDS: bounds + window do not fit; exhaustive kernel does
            use_sea = False
        else:
            _check(st, "x265amd_me_fullsearch_sea")
    if not use_sea:
        _check(_lib().x265amd_me_fullsearch(cp, ref.data_ptr(), w, h, cs, ctu, search_range, f, cfs,
                                            ref.stride(0) if f > 1 else 0, nrefs, ref.stride(1) if nrefs > 1 else 0,
                                            out.data_ptr(), _stream()), "x265amd_me_fullsearch")
    return out


def me_fullsearch_pu(cur, ref, search_range=64, sizes=(64, 32, 16)):
    """Every square PU size of every CTU in one pass (x265amd_me_fullsearch_pu).  cur / ref as for
    me_fullsearch.  Returns {size: int32 [F][R][nblocks][2]} with exactly me_fullsearch(ctu=size)'s content."""
    cp, f, h, w, cs, cfs = _planes(cur, torch.uint8, "cur")
    _dev(ref, torch.uint8, "ref")
    if ref.dim() == 2:
        ref = ref.unsqueeze(0)
    if ref.dim() == 3:
        ref = ref.unsqueeze(1)
    if ref.dim() != 4 or ref.shape[0] != f or ref.shape[2:] != (h, w) or ref.stride(3) != 1 or ref.stride(2) != cs:
        raise ValueError("ref: need [F][R][H][W] with the row stride of cur")
    nrefs = ref.shape[1]
    if any(b not in (64, 32, 16) for b in sizes):
        raise ValueError("sizes: 64, 32, 16")
    out = {b: torch.empty((f, nrefs, max(h // b, 0) * max(w // b, 0), 2), dtype=torch.int32, device=cur.device) for b in sizes}
    ptr = lambda b: out[b].data_ptr() if b in out else None
    _check(_lib().x265amd_me_fullsearch_pu(cp, ref.data_ptr(), w, h, cs, search_range, f, cfs,
                                           ref.stride(0) if f > 1 else 0, nrefs, ref.stride(1) if nrefs > 1 else 0,
                                           ptr(64), ptr(32), ptr(16), _stream()), "x265amd_me_fullsearch_pu")
    return out


# include/x265_amd.h: enum x265amd_pu_shape, (width, height) in that order
PU_SHAPES = ((64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16), (8, 8))


def me_fullsearch_pu_all(cur, ref, search_range=64, shapes=PU_SHAPES):
    """Every inter PU shape of every CTU in one pass (x265amd_me_fullsearch_pu_all): square 64 .. 8 and the 2NxN /
    Nx2N rectangles.  cur / ref as for me_fullsearch.  Returns {(w, h): int32 [F][R][H / h][W / w][2]}: the
    exhaustive search of the w x h blocks tiling the picture, each under its own legality."""
    cp, f, h, w, cs, cfs = _planes(cur, torch.uint8, "cur")
    _dev(ref, torch.uint8, "ref")
    if ref.dim() == 2:
        ref = ref.unsqueeze(0)
    if ref.dim() == 3:
        ref = ref.unsqueeze(1)
    if ref.dim() != 4 or ref.shape[0] != f or ref.shape[2:] != (h, w) or ref.stride(3) != 1 or ref.stride(2) != cs:
        raise ValueError("ref: need [F][R][H][W] with the row stride of cur")
    nrefs = ref.shape[1]
    shapes = tuple(tuple(s) for s in shapes)
    if not shapes or any(s not in PU_SHAPES for s in shapes):
        raise ValueError(f"shapes: some of {PU_SHAPES}")
    out = {s: torch.empty((f, nrefs, h // s[1], w // s[0], 2), dtype=torch.int32, device=cur.device) for s in shapes}
    ptrs = (C.c_void_p * len(PU_SHAPES))(*[out[s].data_ptr() if s in out else None for s in PU_SHAPES])
    _check(_lib().x265amd_me_fullsearch_pu_all(cp, ref.data_ptr(), w, h, cs, search_range, f, cfs,
                                               ref.stride(0) if f > 1 else 0, nrefs, ref.stride(1) if nrefs > 1 else 0,
                                               ptrs, _stream()), "x265amd_me_fullsearch_pu_all")
    return out


def intra_decide_luma(src, xy, size, ref, mpm, lam, want_sads=False, want_pred=False):
    """The 35-mode luma intra decision of a batch of independent CUs (x265amd_intra_decide_luma).
    src: uint8 [H][W] plane; xy: int32 [n][2] (x, y) corners; ref: uint8 [n][>= 4 size + 1] padded reference
    samples (pucPixRef[0]); mpm: uint8 [n][3]; lam: lambda (the reference uses QP).
    Returns (best int32 [n][2] = (mode, cost), sads int32 [n][35] or None, pred uint8 [n][size][size] or None)."""
    _dev(src, torch.uint8, "src")
    _dev(xy, torch.int32, "xy")
    _dev(ref, torch.uint8, "ref")
    _dev(mpm, torch.uint8, "mpm")
    if src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("src: need a [H][W] plane with unit pixel stride")
    n = xy.shape[0]
    if xy.shape != (n, 2) or not xy.is_contiguous() or mpm.shape != (n, 3) or not mpm.is_contiguous():
        raise ValueError("xy: contiguous [n][2]; mpm: contiguous [n][3]")
    if ref.dim() != 2 or ref.shape[0] != n or ref.stride(1) != 1 or ref.shape[1] < 4 * size + 1:
        raise ValueError("ref: need [n][>= 4 * size + 1]")
    best = torch.empty((n, 2), dtype=torch.int32, device=src.device)
    sads = torch.empty((n, 35), dtype=torch.int32, device=src.device) if want_sads else None
    pred = torch.empty((n, size, size), dtype=torch.uint8, device=src.device) if want_pred else None
    _check(_lib().x265amd_intra_decide_luma(src.data_ptr(), src.stride(0), src.shape[1], src.shape[0], xy.data_ptr(), n,
                                            size, ref.data_ptr(), ref.stride(0) if n > 1 else max(ref.shape[1], 4 * size + 1),
                                            mpm.data_ptr(), lam, best.data_ptr(),
                                            sads.data_ptr() if want_sads else None, pred.data_ptr() if want_pred else None,
                                            _stream()), "x265amd_intra_decide_luma")
    return best, sads, pred


def _check_out(out, spec, device):
    """the library writes through these pointers: a buffer of another size, type, layout or device would be written
    past its end, so a caller's `out` is checked field by field before anything is launched"""
    if set(out) != set(spec):
        raise ValueError(f"out: need exactly the buffers {sorted(spec)}")
    for k, (shape, dtype) in spec.items():
        t = out[k]
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or t.device != device:
            raise ValueError(f"out[{k!r}]: need a contiguous {dtype} tensor of shape {tuple(shape)} on {device}")


class IntraFrame(C.Structure):
    """include/x265_amd.h: x265amd_intra_frame"""
    _fields_ = ([("width", C.c_int32), ("height", C.c_int32), ("cu", C.c_int32), ("qp", C.c_int32),
                 ("stride_y", C.c_int32), ("stride_c", C.c_int32)] +
                [(n, C.c_void_p) for n in ("d_src_y", "d_src_u", "d_src_v", "d_rec_y", "d_rec_u", "d_rec_v",
                                           "d_pred_y", "d_pred_u", "d_pred_v", "d_level_y", "d_level_u", "d_level_v",
                                           "d_mode_y", "d_mode_c", "d_cost_y", "d_cost_c",
                                           "d_abs_sum_y", "d_abs_sum_u", "d_abs_sum_v")])


def intra_frame_buffers(h, w, cu, device):
    """The output planes of intra_encode_frame.  U and V are the two halves of one tensor each (as in a 4:2:0 frame
    buffer), which lets the library run their transform stage as one batched launch."""
    ncus = (h // cu) * (w // cu) if cu in (16, 32) else 0
    z8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device=device)
    rec_c, pred_c = z8(2, h // 2, w // 2), z8(2, h // 2, w // 2)
    level_c = torch.zeros((2, h // 2, w // 2), dtype=torch.int16, device=device)
    sum_c = torch.zeros((2, ncus), dtype=torch.int32, device=device)
    zi = lambda: torch.zeros(ncus, dtype=torch.int32, device=device)
    return {"rec_y": z8(h, w), "rec_u": rec_c[0], "rec_v": rec_c[1], "pred_y": z8(h, w), "pred_u": pred_c[0], "pred_v": pred_c[1],
            "level_y": torch.zeros((h, w), dtype=torch.int16, device=device), "level_u": level_c[0], "level_v": level_c[1],
            "mode_y": z8(ncus), "mode_c": z8(ncus), "cost_y": zi(), "cost_c": zi(),
            "abs_sum_y": zi(), "abs_sum_u": sum_c[0], "abs_sum_v": sum_c[1]}


def intra_encode_frame(y, u, v, cu, qp, wave_by_wave=False, out=None):
    """One intra frame on the GPU (x265amd_intra_encode_frame): the pixel path of the reference's encode loop.
    y: uint8 [H][W]; u, v: uint8 [H/2][W/2] (contiguous).  Returns a dict: rec_y / rec_u / rec_v, mode_y / mode_c
    (uint8 [ncus]), cost_y / cost_c (int32 [ncus]), level_y / level_u / level_v (int16 planes), abs_sum_y / _u / _v.
    wave_by_wave: drive the anti-diagonals from here (x265amd_intra_predict_wave + x265amd_tq_roundtrip per plane)
    instead of the library's own loop -- the same launches.  out: buffers of an earlier call / intra_frame_buffers
    to reuse (no allocation in the call: it can then be captured into a graph)."""
    for t, n in ((y, "y"), (u, "u"), (v, "v")):
        _dev(t, torch.uint8, n)
        if t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"{n}: need a contiguous [H][W] plane")
    h, w = y.shape
    if u.shape != (h // 2, w // 2) or v.shape != u.shape:
        raise ValueError("u, v: need [H/2][W/2]")
    ncus = (h // cu) * (w // cu) if cu in (16, 32) else 0
    dv = y.device
    o = out if out is not None else intra_frame_buffers(h, w, cu, dv)
    hc, wc = h // 2, w // 2
    _check_out(o, {"rec_y": ((h, w), torch.uint8), "rec_u": ((hc, wc), torch.uint8), "rec_v": ((hc, wc), torch.uint8),
                   "pred_y": ((h, w), torch.uint8), "pred_u": ((hc, wc), torch.uint8), "pred_v": ((hc, wc), torch.uint8),
                   "level_y": ((h, w), torch.int16), "level_u": ((hc, wc), torch.int16), "level_v": ((hc, wc), torch.int16),
                   "mode_y": ((ncus,), torch.uint8), "mode_c": ((ncus,), torch.uint8),
                   "cost_y": ((ncus,), torch.int32), "cost_c": ((ncus,), torch.int32), "abs_sum_y": ((ncus,), torch.int32),
                   "abs_sum_u": ((ncus,), torch.int32), "abs_sum_v": ((ncus,), torch.int32)}, dv)
    f = IntraFrame()
    f.width, f.height, f.cu, f.qp, f.stride_y, f.stride_c = w, h, cu, qp, w, w // 2
    f.d_src_y, f.d_src_u, f.d_src_v = y.data_ptr(), u.data_ptr(), v.data_ptr()
    for k, t in o.items():
        setattr(f, "d_" + k, t.data_ptr())
    lib = _lib()
    if not wave_by_wave:
        _check(lib.x265amd_intra_encode_frame(C.byref(f), _stream()), "x265amd_intra_encode_frame")
    else:
        nw = lib.x265amd_intra_waves(C.byref(f))
        if nw < 0:
            raise X265AmdError("x265amd_intra_waves: bad frame descriptor")
        for d in range(nw):
            _check(lib.x265amd_intra_predict_wave(C.byref(f), d, _stream()), "x265amd_intra_predict_wave")
            for pl, tu_, q_ in (("y", cu, qp), ("u", cu // 2, chroma_qp(qp)), ("v", cu // 2, chroma_qp(qp))):
                tq_roundtrip({"y": y, "u": u, "v": v}[pl], o["pred_" + pl], tu_, q_, SLICE_I,
                             out=(o["level_" + pl], o["abs_sum_" + pl], o["rec_" + pl]))
    return o


def intra_encode_frames(yuv, width, height, cu, qp, out=None):
    """A batch of independent intra frames in one pass over the anti-diagonals (x265amd_intra_encode_frames).
    yuv: uint8 [N][W*H*3/2] on the device, 4:2:0 frames Y,U,V one after the other (the reference's input format).
    Returns a dict of batched outputs in the same layout: rec, pred (uint8 [N][W*H*3/2]), level (int16
    [N][W*H*3/2]), mode_y / mode_c (uint8 [N][ncus]), cost_y / cost_c, abs_sum_y / _u / _v (int32 [N][ncus]).
    out: the dict of an earlier call to reuse (nothing is allocated then)."""
    _dev(yuv, torch.uint8, "yuv")
    w, h = width, height
    fsz = w * h * 3 // 2
    if yuv.dim() != 2 or yuv.shape[1] != fsz or not yuv.is_contiguous():
        raise ValueError("yuv: need a contiguous [N][W*H*3/2] tensor")
    n = yuv.shape[0]
    ncus = (h // cu) * (w // cu) if cu in (16, 32) else 0
    dv = yuv.device
    if out is None:
        zi = lambda: torch.zeros((n, ncus), dtype=torch.int32, device=dv)
        out = {"rec": torch.zeros((n, fsz), dtype=torch.uint8, device=dv), "pred": torch.zeros((n, fsz), dtype=torch.uint8, device=dv),
               "level": torch.zeros((n, fsz), dtype=torch.int16, device=dv),
               "mode_y": torch.zeros((n, ncus), dtype=torch.uint8, device=dv), "mode_c": torch.zeros((n, ncus), dtype=torch.uint8, device=dv),
               "cost_y": zi(), "cost_c": zi(), "abs_sum_y": zi(), "abs_sum_u": zi(), "abs_sum_v": zi()}
    _check_out(out, {"rec": ((n, fsz), torch.uint8), "pred": ((n, fsz), torch.uint8), "level": ((n, fsz), torch.int16),
                     "mode_y": ((n, ncus), torch.uint8), "mode_c": ((n, ncus), torch.uint8),
                     "cost_y": ((n, ncus), torch.int32), "cost_c": ((n, ncus), torch.int32),
                     "abs_sum_y": ((n, ncus), torch.int32), "abs_sum_u": ((n, ncus), torch.int32),
                     "abs_sum_v": ((n, ncus), torch.int32)}, dv)
    f = IntraFrame()
    f.width, f.height, f.cu, f.qp, f.stride_y, f.stride_c = w, h, cu, qp, w, w // 2
    ysz, csz = w * h, w * h // 4
    for name, t, esz in (("src", yuv, 1), ("rec", out["rec"], 1), ("pred", out["pred"], 1), ("level", out["level"], 2)):
        base = t.data_ptr()
        setattr(f, f"d_{name}_y", base)
        setattr(f, f"d_{name}_u", base + esz * ysz)
        setattr(f, f"d_{name}_v", base + esz * (ysz + csz))
    for k in ("mode_y", "mode_c", "cost_y", "cost_c", "abs_sum_y", "abs_sum_u", "abs_sum_v"):
        setattr(f, "d_" + k, out[k].data_ptr())
    _check(_lib().x265amd_intra_encode_frames(C.byref(f), n, fsz, fsz, ncus, _stream()), "x265amd_intra_encode_frames")
    return out


class IntraSyntax(C.Structure):
    """include/x265_amd.h: x265amd_intra_syntax"""
    _fields_ = ([(n, C.c_int32) for n in ("width", "height", "cu", "qp", "poc", "stride_y", "stride_c")] +
                [(n, C.c_void_p) for n in ("level_y", "level_u", "level_v", "mode_y", "mode_c")])


def intra_write_picture(level_y, level_u, level_v, mode_y, mode_c, cu, qp, poc):
    """The bitstream of one intra picture (x265amd_intra_write_picture; host code): parameter sets, slice header, every
    CU's syntax through CABAC, slice end -- the reference's bytes for these levels and modes.  Arguments are HOST
    arrays (numpy, or CPU / CUDA tensors, which are brought over): int16 level planes [H][W], [H/2][W/2] x 2; uint8
    modes [ncus] (mode_c = the real chroma mode).  Returns a uint8 numpy array."""
    def host(a, dt):
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(a, dtype=dt)
    ly, lu, lv = host(level_y, np.int16), host(level_u, np.int16), host(level_v, np.int16)
    my, mc = host(mode_y, np.uint8).ravel(), host(mode_c, np.uint8).ravel()
    if ly.ndim != 2 or lu.ndim != 2 or lu.shape != lv.shape or lu.shape != (ly.shape[0] // 2, ly.shape[1] // 2):
        raise ValueError("levels: need [H][W], [H/2][W/2], [H/2][W/2]")
    h, w = ly.shape
    ncus = (h // cu) * (w // cu) if cu in (16, 32) else 0
    if my.size != ncus or mc.size != ncus:
        raise ValueError("modes: need one luma and one chroma mode per CU")
    d = IntraSyntax()
    d.width, d.height, d.cu, d.qp, d.poc, d.stride_y, d.stride_c = w, h, cu, qp, poc, w, w // 2
    d.level_y, d.level_u, d.level_v = ly.ctypes.data, lu.ctypes.data, lv.ctypes.data
    d.mode_y, d.mode_c = my.ctypes.data, mc.ctypes.data
    lib = _lib()
    out = np.empty(lib.x265amd_intra_stream_bound(w, h), np.uint8)
    n = C.c_size_t(0)
    _check(lib.x265amd_intra_write_picture(C.byref(d), out.ctypes.data, out.size, C.byref(n)), "x265amd_intra_write_picture")
    return out[:n.value].copy()


def mv_to_numpy(mv):
    """int32 [...][2] device tensor -> structured numpy array (mvx, mvy, sad)."""
    a = mv.detach().cpu().numpy()
    return np.ascontiguousarray(a).view(MV_DTYPE).reshape(a.shape[:-1])


# --------------------------------------------------------------------------
# xSubDct + xQuant / xDeQuant + xIDctAdd (inc/x265.h:295-298)
# --------------------------------------------------------------------------
def _tq_desc(tu, qp, slice_type, src=None, pred=None, mv=None, mv_ctu=64, modes=None, level=None, coef=None,
             abs_sum=None, rec=None, pred_origin=(0, 0), mv_shift=0, check_mv=False):
    d = TqDesc()
    d.flags = TQ_CHECK_MV if check_mv else 0
    geom = None
    if src is not None:
        d.d_src, f, h, w, d.src_stride, d.src_frame_stride = _planes(src, torch.uint8, "src")
        geom = (f, h, w)
    if level is not None:
        d.d_level, f, h, w, d.level_stride, d.level_frame_stride = _planes(level, torch.int16, "level")
        if geom is not None and geom != (f, h, w):
            raise ValueError("level / src geometry mismatch")
        geom = (f, h, w)
    if geom is None:
        raise ValueError("need src or level")
    f, h, w = geom
    d.width, d.height, d.tu, d.qp, d.slice_type, d.nframes = w, h, tu, qp, slice_type, f
    pp, pf, ph, pw, d.pred_stride, d.pred_frame_stride = _planes(pred, torch.uint8, "pred")
    if pf != f:
        raise ValueError("pred / src frame count mismatch")
    oy, ox = pred_origin
    if mv is None and (ph - oy < h or pw - ox < w):
        raise ValueError("pred plane smaller than the picture")
    d.d_pred = pp + oy * d.pred_stride + ox
    if mv is not None:
        _dev(mv, torch.int32, "mv")
        nctu = (h // mv_ctu) * (w // mv_ctu)
        if mv.numel() != f * nctu * 2 or not mv.is_contiguous():
            raise ValueError("mv: need contiguous int32 [F][nctu][2]")
        d.d_mv, d.mv_ctu, d.mv_frame_stride = mv.data_ptr(), mv_ctu, nctu
        d.mv_shift = mv_shift
    if modes is not None:
        _dev(modes, torch.uint8, "modes")
        ntu = (h // tu) * (w // tu)
        if modes.numel() != f * ntu or not modes.is_contiguous():
            raise ValueError("modes: need contiguous uint8 [F][ntu]")
        d.d_modes, d.modes_frame_stride = modes.data_ptr(), ntu
    if coef is not None:
        cp, cf, chh, cw, cstr, cfs = _planes(coef, torch.int16, "coef")
        if (cf, chh, cw, cstr, cfs) != (f, h, w, d.level_stride, d.level_frame_stride):
            raise ValueError("coef must have the layout of level")
        d.d_coef = cp
    if abs_sum is not None:
        _dev(abs_sum, torch.int32, "abs_sum")
        if tu > 0 and (not abs_sum.is_contiguous() or abs_sum.numel() < f * (h // tu) * (w // tu)):
            raise ValueError("abs_sum: need a contiguous int32 tensor of at least [F][TUs per plane] elements")
        d.d_abs_sum, d.abs_sum_frame_stride = abs_sum.data_ptr(), (h // tu) * (w // tu)
    if rec is not None:
        d.d_rec, rf, rh, rw, d.rec_stride, d.rec_frame_stride = _planes(rec, torch.uint8, "rec")
        if (rf, rh, rw) != (f, h, w):
            raise ValueError("rec / picture geometry mismatch")
    return d


def _alloc_like(src, dtype, tu=None):
    s = src if src.dim() == 3 else src.unsqueeze(0)
    f, h, w = s.shape
    if tu is None:
        return torch.empty((f, h, w), dtype=dtype, device=src.device)
    return torch.empty((f, (h // tu) * (w // tu)), dtype=dtype, device=src.device)


def tq_forward(src, pred, tu, qp, slice_type=SLICE_I, mv=None, mv_ctu=64, modes=None, want_coef=False,
               pred_origin=(0, 0), mv_shift=0, check_mv=False):
    """xSubDct + xQuant over whole planes.  Returns (level, abs_sum[, coef])."""
    level = _alloc_like(src, torch.int16)
    sums = _alloc_like(src, torch.int32, tu if tu in (4, 8, 16, 32) else 4)
    coef = _alloc_like(src, torch.int16) if want_coef else None
    d = _tq_desc(tu, qp, slice_type, src=src, pred=pred, mv=mv, mv_ctu=mv_ctu, modes=modes, level=level,
                 coef=coef, abs_sum=sums, pred_origin=pred_origin, mv_shift=mv_shift, check_mv=check_mv)
    _check(_lib().x265amd_tq_forward(C.byref(d), _stream()), "x265amd_tq_forward")
    return (level, sums, coef) if want_coef else (level, sums)


def tq_inverse(level, pred, tu, qp, slice_type=SLICE_I, mv=None, mv_ctu=64, modes=None, pred_origin=(0, 0),
               mv_shift=0):
    """xDeQuant + xIDctAdd over whole planes.  Returns rec."""
    rec = _alloc_like(level, torch.uint8)
    d = _tq_desc(tu, qp, slice_type, pred=pred, mv=mv, mv_ctu=mv_ctu, modes=modes, level=level, rec=rec,
                 pred_origin=pred_origin, mv_shift=mv_shift)
    _check(_lib().x265amd_tq_inverse(C.byref(d), _stream()), "x265amd_tq_inverse")
    return rec


def idct_add(coef, pred, tu, modes=None, pred_origin=(0, 0)):
    """xIDctAdd alone over whole planes: `coef` holds dequantised coefficients.  Returns rec."""
    rec = _alloc_like(coef, torch.uint8)
    d = _tq_desc(tu, 0, SLICE_I, pred=pred, modes=modes, level=coef, rec=rec, pred_origin=pred_origin)
    _check(_lib().x265amd_idct_add(C.byref(d), _stream()), "x265amd_idct_add")
    return rec


def tq_roundtrip(src, pred, tu, qp, slice_type=SLICE_I, mv=None, mv_ctu=64, modes=None, pred_origin=(0, 0),
                 out=None, mv_shift=0, check_mv=False):
    """Fused forward + inverse (src/encode.cpp:863-925).  Returns (level, abs_sum, rec).
    `out` = (level, abs_sum, rec) preallocated tensors to reuse (no allocation in the call)."""
    if out is None:
        level = _alloc_like(src, torch.int16)
        sums = _alloc_like(src, torch.int32, tu if tu in (4, 8, 16, 32) else 4)
        rec = _alloc_like(src, torch.uint8)
    else:
        level, sums, rec = out
    d = _tq_desc(tu, qp, slice_type, src=src, pred=pred, mv=mv, mv_ctu=mv_ctu, modes=modes, level=level,
                 abs_sum=sums, rec=rec, pred_origin=pred_origin, mv_shift=mv_shift, check_mv=check_mv)
    _check(_lib().x265amd_tq_roundtrip(C.byref(d), _stream()), "x265amd_tq_roundtrip")
    return level, sums, rec


def quant_plane(coef, tu, qp, slice_type=SLICE_I):
    """xQuant on a coefficient plane [H][W] -> (level, abs_sum)."""
    _dev(coef, torch.int16, "coef")
    h, w = coef.shape
    level = torch.empty_like(coef)
    sums = torch.empty(((h // tu) * (w // tu)) if tu > 0 else 0, dtype=torch.int32, device=coef.device)
    _check(_lib().x265amd_quant_plane(coef.data_ptr(), level.data_ptr(), coef.stride(0), w, h, tu, qp, slice_type,
                                      sums.data_ptr(), _stream()), "x265amd_quant_plane")
    return level, sums


def dequant_plane(level, tu, qp):
    """xDeQuant on a level plane [H][W] -> coef."""
    _dev(level, torch.int16, "level")
    h, w = level.shape
    coef = torch.empty_like(level)
    _check(_lib().x265amd_dequant_plane(level.data_ptr(), coef.data_ptr(), level.stride(0), w, h, tu, qp, _stream()),
           "x265amd_dequant_plane")
    return coef
