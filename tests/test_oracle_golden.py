"""The CPU restatement (oracle/) against the committed golden vectors that
tests/golden/gen_golden.py produced from the compiled reference.  CPU only."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN, load_golden
from oracle import pyoracle


def test_sad_golden(oracle):
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "sad"]
    assert len(cases) == 30
    for m in cases:
        k = m["id"]
        got = oracle.sad(m["w"], m["n"], z[f"sad{k}_src"], m["ss"], z[f"sad{k}_ref"], m["sr"])
        assert got == int(z[f"sad{k}_val"]), m


def test_transform_1d_golden(oracle):
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "t1d"]
    assert len(cases) == 20
    for m in cases:
        k = m["id"]
        f = oracle.fwd_1d if m["dir"] == "fwd" else oracle.inv_1d
        got = f(m["idx"], z[f"t1d{k}_in"], 32, m["n"], m["shift"])
        n = m["n"]
        np.testing.assert_array_equal(got.reshape(n, 32)[:, :n], z[f"t1d{k}_out"].reshape(n, 32)[:, :n], str(m))


def test_tq_chain_golden(oracle):
    """residual -> transform -> quant -> dequant -> inverse -> recon, stage by stage
    (the reference's own CHECK_TV idea, src/pixel.cpp:619-631,653-665,709-721,742-775)."""
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "tq"]
    assert len(cases) > 200
    for m in cases:
        k, n = m["id"], m["n"]
        v = lambda a: a.reshape(n, 32)[:, :n]
        src, pred = z[f"tq{k}_src"], z[f"tq{k}_pred"]
        coef = oracle.sub_dct(src, pred, 32, n, n, m["mode"])
        np.testing.assert_array_equal(v(coef), v(z[f"tq{k}_coef"]), f"coef {m}")
        level, s = oracle.quant(coef, 32, m["qp"], n, n, m["slice"])
        np.testing.assert_array_equal(v(level), v(z[f"tq{k}_level"]), f"level {m}")
        assert s == int(z[f"tq{k}_sum"]), m
        deq = oracle.dequant(level, 32, m["qp"], n, n, m["slice"])
        np.testing.assert_array_equal(v(deq), v(z[f"tq{k}_deq"]), f"deq {m}")
        rec = oracle.idct_add(deq, pred, 32, n, n, m["mode"])
        np.testing.assert_array_equal(v(rec), v(z[f"tq{k}_rec"]), f"rec {m}")


def test_me_golden(oracle):
    z, meta = load_golden("me_small.npz")
    for m in meta:
        k = m["id"]
        got = oracle.me_fullsearch(z[f"me{k}_cur"], z[f"me{k}_ref"], m["ctu"], m["range"], nthreads=4)
        want = z[f"me{k}_mv"]
        for f in ("mvx", "mvy", "sad"):
            np.testing.assert_array_equal(got[f], want[f], f"{f} {m}")
    # the all-ties case resolves to the first candidate in raster order that is inside the picture
    mv = z["me2_mv"]
    assert mv["sad"].tolist() == [0, 0, 0, 0]
    assert (mv["mvx"][0], mv["mvy"][0]) == (0, 0)      # CTU (0,0): dy<0 and dx<0 leave the picture
    assert (mv["mvx"][3], mv["mvy"][3]) == (-7, -7)    # CTU (1,1): top-left candidate is legal


def test_tq_frame_golden(oracle):
    z, meta = load_golden("tq_frame.npz")
    cur, refp, mv = z["cur"], z["refp"], z["mv"]
    h, w = cur.shape
    for m in meta:
        k, n = m["id"], m["n"]
        if m["mv"]:
            # pred = pointer into the larger plane: origin moved into the margin, stride stays
            level, sums = _fwd_strided(oracle, cur, refp, 32, 32, mv, n, m["qp"], m["slice"])
            rec = _inv_strided(oracle, level, refp, 32, 32, mv, n, m["qp"], m["slice"])
        else:
            pred = np.ascontiguousarray(refp[32:32 + h, 32:32 + w])
            level, sums = oracle.tq_forward(cur, pred, n, m["qp"], m["slice"])
            rec = oracle.tq_inverse(level, pred, n, m["qp"], m["slice"])
        np.testing.assert_array_equal(level, z[f"f{k}_level"], str(m))
        np.testing.assert_array_equal(sums, z[f"f{k}_sums"], str(m))
        np.testing.assert_array_equal(rec, z[f"f{k}_rec"], str(m))


def _fwd_strided(oracle, cur, refp, ox, oy, mv, n, qp, st):
    """tq_forward with pred = pointer into a larger plane (origin offset, full stride)."""
    import ctypes as C
    h, w = cur.shape
    level = np.zeros((h, w), np.int16)
    sums = np.zeros((h // n) * (w // n), np.uint32)
    base = refp.ctypes.data + oy * refp.shape[1] + ox
    oracle.lib.orc_tq_forward(cur.ctypes.data_as(C.POINTER(C.c_uint8)), w,
                              C.cast(base, C.POINTER(C.c_uint8)), refp.shape[1], mv.ctypes.data, 64,
                              w, h, n, qp, st, None, level.ctypes.data_as(C.POINTER(C.c_int16)), w,
                              sums.ctypes.data_as(C.POINTER(C.c_uint32)), 1)
    return level, sums


def _inv_strided(oracle, level, refp, ox, oy, mv, n, qp, st):
    import ctypes as C
    h, w = level.shape
    rec = np.zeros((h, w), np.uint8)
    base = refp.ctypes.data + oy * refp.shape[1] + ox
    oracle.lib.orc_tq_inverse(level.ctypes.data_as(C.POINTER(C.c_int16)), w,
                              C.cast(base, C.POINTER(C.c_uint8)), refp.shape[1], mv.ctypes.data, 64,
                              w, h, n, qp, st, None, rec.ctypes.data_as(C.POINTER(C.c_uint8)), w, 1)
    return rec


def test_intra_cfg1_fixture_is_the_surveyed_run():
    """BASELINE.json configs[0] (plumbing, CPU only): the committed fixture is the
    reference encoder's output on the SURVEY App. B probe; the input generator
    must still reproduce the exact input those MD5s belong to."""
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    yuv = synth.make_intra_probe(doc["width"], doc["height"], doc["frames"])
    assert hashlib.md5(yuv.tobytes()).hexdigest() == doc["input_md5"]
    assert doc["recon_md5"] == doc["recon_md5_hashlib"]          # src/md5.cpp == RFC 1321
    assert doc["recon_md5"][0] == "b254f8bfc75464d975b9e488a1a99383"
    assert doc["recon_md5"][7] == "07ea1133114f3e594a7ea24488a198e9"


def test_intra_cfg1_live(reference):
    """Where the reference can be compiled (this container), re-run configs[0] and
    compare with the fixture: identical bitstream and recon MD5 for all 8 frames."""
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    yuv = synth.make_intra_probe(doc["width"], doc["height"], doc["frames"])
    rec, nbytes, md5, bs = reference.encode_intra(yuv, doc["width"], doc["height"], doc["cu"], doc["qp"])
    assert [int(x) for x in nbytes] == doc["bytes_per_frame"]
    assert hashlib.md5(bs.tobytes()).hexdigest() == doc["bitstream_md5"]
    assert [bytes(m).hex() for m in md5] == doc["recon_md5"]


def test_intra_decide_recorded_from_the_reference(oracle):
    """tests/golden/intra_decide.npz holds the reference encoder's OWN luma decisions, recorded from inside
    xEncEncode on BASELINE.json configs[0] (frames 0 and 5) and on a CU-32 probe: per CU the reference samples,
    their smoothed copy, the most probable modes, the SAD and a hash of the prediction of each of the 35 modes,
    and the mode kept.  The restatement must reproduce all of it from (source, samples, most probable modes)."""
    from oracle.pyoracle import fnv1a
    from x265_amd import synth
    z, meta = load_golden("intra_decide.npz")
    for m in meta:
        k, n = m["key"], m["cu"]
        yuv = synth.make_intra_probe(m["width"], m["height"], m["frame"] + 1)
        assert hashlib.md5(yuv[m["frame"]].tobytes()).hexdigest() == m["input_md5"]
        Y = yuv[m["frame"], :m["width"] * m["height"]].reshape(m["height"], m["width"])
        xy, ref0, ref1, mpm = z[k + "_xy"], z[k + "_ref0"], z[k + "_ref1"], z[k + "_mpm"]
        assert len(xy) == m["ncus"] == (m["width"] // n) * (m["height"] // n)
        lam = m["qp"]                                        # src/encode.cpp:597
        for i in range(m["ncus"]):
            x, y = int(xy[i, 0]), int(xy[i, 1])
            src = np.ascontiguousarray(Y[y:y + n, x:x + n])
            r0 = np.ascontiguousarray(ref0[i])
            r1 = oracle.intra_filter(r0)
            np.testing.assert_array_equal(r1, ref1[i], f"{k} CU {i}: smoothed samples")
            sads, bm, bc, pred = oracle.intra_decide_luma(src, r0, np.ascontiguousarray(mpm[i]), lam)
            np.testing.assert_array_equal(sads, z[k + "_sad"][i], f"{k} CU {i}: SADs")
            assert bm == z[k + "_best"][i], (k, i)
            w = 1 if bm == mpm[i, 0] else 2 if bm in (mpm[i, 1], mpm[i, 2]) else 3
            assert bc == sads[bm] + w * lam
            if i % 7 == 0:                                   # every mode's prediction, on a sample of the CUs
                for mode in range(35):
                    assert fnv1a(oracle.intra_pred_luma(r0, r1, n, mode)) == z[k + "_fnv"][i, mode], (k, i, mode)
            assert fnv1a(pred) == z[k + "_fnv"][i, bm]


def test_intra_frame_restatement_reproduces_cfg1_md5(oracle):
    """BASELINE.json configs[0] without the reference: the restated pixel path of the intra encode loop
    (orc_intra_encode_frame: neighbour samples, luma and chroma decisions, transform stage, reconstruction) gives,
    frame by frame, the reconstruction whose MD5 the reference printed (tests/golden/intra_cfg1.json), and on
    frames 0 and 5 the luma modes recorded from inside the reference (intra_decide.npz)."""
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    z, meta = load_golden("intra_decide.npz")
    yuv = synth.make_intra_probe(doc["width"], doc["height"], doc["frames"])
    for fr in range(doc["frames"]):
        rec, my, mc, _ = oracle.intra_encode_frame(yuv[fr], doc["width"], doc["height"], doc["cu"], doc["qp"])
        assert hashlib.md5(rec.tobytes()).hexdigest() == doc["recon_md5"][fr], fr
        for m in meta:
            if (m["width"], m["height"], m["cu"], m["qp"], m["frame"]) == (doc["width"], doc["height"], doc["cu"], doc["qp"], fr):
                np.testing.assert_array_equal(my, z[m["key"] + "_best"])
