"""x265amd_intra_write_picture -- the host-side syntax writer of an intra picture (x265_amd/csrc/intra_stream.hip:
parameter sets, slice header, xWriteCU, residual coding, CABAC) -- against the reference's bytes.  Host code: these
tests need no GPU.  Pins: tests/golden/intra_stream.npz (the compiled reference's own writer on random syntax, and a
CU-32 encode of the reference) and tests/golden/intra_cfg1.json (BASELINE configs[0]); with oracle/_ref present,
fresh random syntax against the reference's writer as well."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import x265_amd
from conftest import GOLDEN, load_golden
from x265_amd import api, synth


def _planes(lv, w, h):
    return lv[:w * h].reshape(h, w), lv[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), lv[w * h * 5 // 4:].reshape(h // 2, w // 2)


def _write(lv, my, mc, w, h, cu, qp):
    """all frames of a case -> (bytes per frame, the stream)"""
    parts = [x265_amd.intra_write_picture(*_planes(lv[f], w, h), my[f], mc[f], cu, qp, f) for f in range(len(lv))]
    return [len(p) for p in parts], np.concatenate(parts)


def test_writer_reproduces_the_reference_writer_on_random_syntax():
    z, meta = load_golden("intra_stream.npz")
    assert len(meta["random"]) == 6
    for m in meta["random"]:
        k = m["key"]
        nbytes, bs = _write(z[k + "_levels"], z[k + "_mode_y"], z[k + "_mode_c"], m["width"], m["height"], m["cu"], m["qp"])
        assert nbytes == m["bytes_per_frame"], k
        assert bs.tobytes() == z[k + "_bits"].tobytes(), k


@pytest.mark.parametrize("which", ["cfg1", "cu32"])
def test_writer_over_restated_pixel_path_gives_the_reference_bitstream(oracle, which):
    """The whole intra encoder without the compiled reference: the restated pixel path (orc_intra_encode_frame, the
    checker of x265amd_intra_encode_frame) feeding the product's writer gives BASELINE configs[0]'s bitstream (CU 16)
    and the CU-32 encode's, byte for byte.  The GPU twin of this test is in test_gpu_parity.py."""
    if which == "cfg1":
        with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
            doc = json.load(f)
    else:
        doc = load_golden("intra_stream.npz")[1]["encode_cu32"]
    w, h, cu, qp, nf = doc["width"], doc["height"], doc["cu"], doc["qp"], doc["frames"]
    yuv = synth.make_intra_probe(w, h, nf)
    assert hashlib.md5(yuv.tobytes()).hexdigest() == doc["input_md5"]
    out = [oracle.intra_encode_frame(yuv[f], w, h, cu, qp, want_levels=True) for f in range(nf)]
    nbytes, bs = _write([o[3] for o in out], [o[1] for o in out], [o[2] for o in out], w, h, cu, qp)
    assert nbytes == doc["bytes_per_frame"]
    assert hashlib.md5(bs.tobytes()).hexdigest() == doc["bitstream_md5"]


def test_writer_against_live_reference_writer(reference):
    """fresh random syntax (the fixture's generator, other seeds and geometries) through ref_write_intra_frames"""
    import sys
    sys.path.insert(0, GOLDEN)
    from gen_golden import random_syntax
    rng = np.random.default_rng(77)
    for (w, h, cu, qp, nf, style) in ((128, 32, 16, 45, 2, "sparse"), (64, 96, 32, 8, 2, "dense"), (160, 16, 16, 26, 3, "tail"),
                                      (32, 64, 32, 30, 2, "ones"), (96, 96, 32, 33, 1, "sparse"), (16, 16, 16, 51, 4, "tail")):
        lv, my, mc = random_syntax(rng, w, h, cu, nf, style)
        zero = np.zeros((nf, w * h * 3 // 2), np.uint8)
        rbytes, rbs = reference.write_intra_frames(zero, w, h, cu, qp, zero, lv, my, mc)
        nbytes, bs = _write(lv, my, mc, w, h, cu, qp)
        assert nbytes == [int(x) for x in rbytes], (w, h, cu, style)
        assert bs.tobytes() == np.asarray(rbs, np.uint8).tobytes(), (w, h, cu, style)


def test_writer_rejects_what_the_reference_cannot_code():
    w, h, cu = 32, 32, 16
    ly, lc = np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16)
    my = np.array([0, 26, 10, 1], np.uint8)
    ok = x265_amd.intra_write_picture(ly, lc, lc, my, my, cu, 30, 0)  # chroma = the luma mode: always a candidate
    assert ok[:5].tobytes() == b"\x00\x00\x00\x01\x47"                 # SPS start code + NAL header, src/encode.cpp:628-629
    with pytest.raises(api.X265AmdError):  # 7 is not among {planar, ver, hor, DC, 34-for-the-duplicate, luma mode}
        x265_amd.intra_write_picture(ly, lc, lc, my, np.array([0, 26, 7, 1], np.uint8), cu, 30, 0)
    with pytest.raises(api.X265AmdError):  # luma mode out of range
        x265_amd.intra_write_picture(ly, lc, lc, np.array([0, 35, 10, 1], np.uint8), my, cu, 30, 0)
    for bad_cu, bad_qp in ((8, 30), (64, 30), (16, 52)):
        with pytest.raises((api.X265AmdError, ValueError)):
            x265_amd.intra_write_picture(ly, lc, lc, my, my, bad_cu, bad_qp, 0)
    # a buffer that is too small: X265AMD_ERR_BUFFER, nothing written past it, the size needed reported
    lib = api._lib()
    d = api.IntraSyntax()
    d.width, d.height, d.cu, d.qp, d.poc, d.stride_y, d.stride_c = w, h, cu, 30, 0, w, w // 2
    d.level_y, d.level_u, d.level_v, d.mode_y, d.mode_c = ly.ctypes.data, lc.ctypes.data, lc.ctypes.data, my.ctypes.data, my.ctypes.data
    buf = np.full(64, 0xAA, np.uint8)
    n = C.c_size_t(0)
    assert lib.x265amd_intra_write_picture(C.byref(d), buf.ctypes.data, 16, C.byref(n)) == -7
    assert n.value == len(ok) and (buf[16:] == 0xAA).all() and buf[:16].tobytes() == ok[:16].tobytes()


@pytest.mark.parametrize("cu", [16, 32])
def test_stream_bound_holds_for_full_range_levels(cu):
    """x265amd_intra_stream_bound (7 bytes per sample): pictures whose every level is +-32767 or -32768 -- the most bins a
    level can cost (Golomb-Rice prefix, Exp-Golomb escape, sign, the context-coded flags) -- and full-range levels at
    random stay under it, at both CU sizes."""
    w, h = 64, 64
    rng = np.random.default_rng(cu)
    ncu = (w // cu) * (h // cu)
    my = rng.integers(0, 35, ncu).astype(np.uint8)
    bound = api._lib().x265amd_intra_stream_bound(w, h)
    assert bound == 256 + 7 * w * h * 3 // 2
    for name, lv in (("+32767", np.full(w * h * 3 // 2, 32767, np.int16)),
                     ("-32768", np.full(w * h * 3 // 2, -32768, np.int16)),
                     ("alternating", np.where(np.arange(w * h * 3 // 2) & 1, 32767, -32768).astype(np.int16)),
                     ("random", rng.integers(-32768, 32768, w * h * 3 // 2).astype(np.int16))):
        bits = x265_amd.intra_write_picture(*_planes(lv, w, h), my, my, cu, 30, 0)
        assert 0 < len(bits) <= bound, (name, len(bits), bound)
        assert len(bits) > w * h * 3 // 2 * 3, name      # such a picture does cost several bytes per sample
