#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference.

Runs only where /root/reference exists (the build container): it drives
oracle/_ref/libx265ref.so, which `make -C oracle ref` compiles from the
reference sources where they lie.  The reference ships no known-answer vectors
of its own (CHEN_TV.TXT, src/encode.cpp:49, is absent), so these files are the
pin for the oracle and, through it, for the HIP kernels.

    python tests/golden/gen_golden.py        # rewrites tests/golden/*.npz, *.json

Everything written is data: inputs and the reference's outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from x265_amd import synth  # noqa: E402

MODE_INVALID = 255


def block_inputs(rng, n, kind):
    """src/pred uint8[n*32] blocks at the reference's scratch stride (MAX_CU_SIZE)."""
    src = np.zeros(n * 32, np.uint8)
    pred = np.zeros(n * 32, np.uint8)
    s = src.reshape(n, 32)
    p = pred.reshape(n, 32)
    if kind == "random":
        s[:, :n] = rng.integers(0, 256, (n, n))
        p[:, :n] = rng.integers(0, 256, (n, n))
    elif kind == "smooth":
        yy, xx = np.mgrid[0:n, 0:n]
        s[:, :n] = (100 + 3 * xx + 2 * yy + rng.integers(0, 8, (n, n))) & 255
        p[:, :n] = (98 + 3 * xx + 2 * yy + rng.integers(0, 4, (n, n))) & 255
    elif kind == "max":  # +255 residual everywhere
        s[:, :n] = 255
    elif kind == "min":  # -255 residual everywhere
        p[:, :n] = 255
    elif kind == "checker":  # +-255 alternating: largest high-frequency energy
        yy, xx = np.mgrid[0:n, 0:n]
        m = ((xx + yy) & 1).astype(bool)
        s[:, :n] = np.where(m, 255, 0)
        p[:, :n] = np.where(m, 0, 255)
    elif kind == "equal":  # zero residual -> CBF 0 path (encode.cpp:901)
        s[:, :n] = rng.integers(0, 256, (n, n))
        p[:] = s
    return src, pred


def gen_primitives(ref):
    rng = np.random.default_rng(20261110)
    out = {}
    meta = []
    # --- SAD: all five widths, several heights, independent strides ---------
    k = 0
    for w in (4, 8, 16, 32, 64):
        for n in (w, max(4, w // 2), 1):
            for kind in ("random", "extreme"):
                ss, sr = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 17))
                a = rng.integers(0, 256, n * ss, dtype=np.uint8)
                b = rng.integers(0, 256, n * sr, dtype=np.uint8)
                if kind == "extreme":
                    a[:] = 255
                    b[:] = 0
                out[f"sad{k}_src"] = a
                out[f"sad{k}_ref"] = b
                out[f"sad{k}_val"] = np.uint32(ref.sad(w, n, a, ss, b, sr))
                meta.append(dict(kind="sad", id=k, w=w, n=n, ss=ss, sr=sr))
                k += 1
    # --- 1-D transforms with their production shifts -------------------------
    k = 0
    for idx, n in ((0, 4), (1, 4), (2, 8), (3, 16), (4, 32)):
        l2 = int(np.log2(n))
        for direction, shifts in (("fwd", (l2 - 1, l2 + 6)), ("inv", (7, 12))):
            for shift in shifts:
                lim = 255 if (direction == "fwd" and shift == l2 - 1) else 32767
                x = np.zeros(n * 32, np.int16)
                x.reshape(n, 32)[:, :n] = rng.integers(-lim, lim + 1, (n, n))
                f = ref.fwd_1d if direction == "fwd" else ref.inv_1d
                y = f(idx, x, 32, n, shift)
                out[f"t1d{k}_in"] = x
                out[f"t1d{k}_out"] = y
                meta.append(dict(kind="t1d", id=k, idx=idx, n=n, dir=direction, shift=shift))
                k += 1
    # --- block TQ chain: residual -> coef -> level -> dequant -> recon ------
    k = 0

    def chain(n, kind, qp, st, mode):
        nonlocal k
        src, pred = block_inputs(rng, n, kind)
        coef = ref.sub_dct(src, pred, 32, n, n, mode)
        level, s = ref.quant(coef, 32, qp, n, n, st)
        deq = ref.dequant(level, 32, qp, n, n, st)
        rec = ref.idct_add(deq, pred, 32, n, n, mode)
        out[f"tq{k}_src"] = src
        out[f"tq{k}_pred"] = pred
        out[f"tq{k}_coef"] = coef
        out[f"tq{k}_level"] = level
        out[f"tq{k}_sum"] = np.uint32(s)
        out[f"tq{k}_deq"] = deq
        out[f"tq{k}_rec"] = rec
        meta.append(dict(kind="tq", id=k, n=n, input=kind, qp=qp, slice=st, mode=mode))
        k += 1

    for n in (4, 8, 16, 32):
        for kind in ("random", "smooth", "max", "min", "checker", "equal"):
            for st in (0, 1):
                chain(n, kind, 32, st, MODE_INVALID)
    # every QP, both roundings, on the two small sizes (cheap) and a sparse sweep on 16/32
    for qp in range(52):
        chain(4, "random", qp, qp & 1, MODE_INVALID)
        chain(8, "smooth", qp, (qp + 1) & 1, MODE_INVALID)
    for qp in (0, 1, 5, 6, 17, 22, 27, 37, 45, 51):
        chain(16, "random", qp, 0, MODE_INVALID)
        chain(32, "random", qp, 1, MODE_INVALID)
        chain(32, "checker", qp, 0, MODE_INVALID)
    # DST selection by intra mode (src/pixel.cpp:678-680, 734-736) -- 4x4 only;
    # larger sizes must ignore the mode
    for mode in (0, 1, 2, 10, 11, 18, 25, 26, 34):
        chain(4, "random", 30, 0, mode)
        chain(4, "smooth", 24, 0, mode)
    for mode in (0, 1, 26):
        chain(8, "random", 30, 0, mode)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)
    print("primitives.npz:", len(meta), "cases")


def gen_me(ref):
    out = {}
    meta = []
    # (w, h, ctu, R): small enough that the reference loop takes seconds
    cases = [(192, 128, 64, 64), (256, 192, 64, 16), (128, 128, 64, 7), (128, 64, 32, 32),
             (64, 64, 16, 16)]
    for k, (w, h, ctu, r) in enumerate(cases):
        seq = synth.make_sequence(w, h, 2, search_range=min(r, 32), seed=777 + k)
        cur, rf = np.ascontiguousarray(seq[1]), np.ascontiguousarray(seq[0])
        if k == 2:  # flat content: every candidate ties -> first in raster order must win
            cur[:] = 90
            rf[:] = 90
        mv = ref.me_fullsearch(cur, rf, ctu, r, nthreads=8)
        out[f"me{k}_cur"] = cur
        out[f"me{k}_ref"] = rf
        out[f"me{k}_mv"] = mv
        meta.append(dict(kind="me", id=k, w=w, h=h, ctu=ctu, range=r))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "me_small.npz"), **out)
    print("me_small.npz:", len(meta), "cases")


def gen_tq_frame(ref):
    out = {}
    meta = []
    w, h = 128, 64
    seq = synth.make_sequence(w + 64, h + 64, 2, search_range=16, seed=4242)
    # a padded reference so that motion-compensated fetches stay inside the plane
    refp = np.ascontiguousarray(seq[0])
    cur = np.ascontiguousarray(seq[1][32:32 + h, 32:32 + w])
    k = 0
    for n in (4, 8, 16, 32):
        for qp, st in ((32, 1), (22, 0)):
            pred = np.ascontiguousarray(refp[32:32 + h, 32:32 + w])
            level, sums, rec = ref.tq_frame(cur, pred, n, qp, st)
            out[f"f{k}_level"], out[f"f{k}_sums"], out[f"f{k}_rec"] = level, sums, rec
            meta.append(dict(kind="tqframe", id=k, n=n, qp=qp, slice=st, mv=False))
            k += 1
    # with per-CTU motion vectors (integer-pel compensated prediction), CTU 64
    mv = np.zeros(2, pyoracle.MV_DTYPE)
    mv["mvx"] = [5, -17]
    mv["mvy"] = [-3, 12]
    for n in (8, 32):
        level, sums, rec = tq_frame_mv(ref, cur, refp, 32, 32, mv, n, 30, 1)
        out[f"f{k}_level"], out[f"f{k}_sums"], out[f"f{k}_rec"] = level, sums, rec
        meta.append(dict(kind="tqframe", id=k, n=n, qp=30, slice=1, mv=True))
        k += 1
    out["cur"], out["refp"], out["mv"] = cur, refp, mv
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "tq_frame.npz"), **out)
    print("tq_frame.npz:", len(meta), "cases")


def tq_frame_mv(ref, cur, refp, ox, oy, mv, n, qp, st):
    """Reference TQ with motion-compensated prediction: materialise the
    prediction plane on the host (block copy per CTU) and run the zero-mv path."""
    h, w = cur.shape
    pred = np.zeros_like(cur)
    ncx = w // 64
    for cy in range(h // 64):
        for cx in range(ncx):
            m = mv[cy * ncx + cx]
            y0, x0 = oy + cy * 64 + int(m["mvy"]), ox + cx * 64 + int(m["mvx"])
            pred[cy * 64:cy * 64 + 64, cx * 64:cx * 64 + 64] = refp[y0:y0 + 64, x0:x0 + 64]
    return ref.tq_frame(cur, pred, n, qp, st)


def gen_intra_cfg1(ref):
    """BASELINE.json configs[0]: 416x240, 8 frames, CU16, QP32, SLICE_I, full xEncEncode."""
    w, h, cu, qp, nf = 416, 240, 16, 32, 8
    yuv = synth.make_intra_probe(w, h, nf)
    rec, nbytes, md5, bs = ref.encode_intra(yuv, w, h, cu, qp)
    doc = dict(width=w, height=h, cu=cu, qp=qp, frames=nf,
               input_md5=hashlib.md5(yuv.tobytes()).hexdigest(),
               bytes_per_frame=[int(x) for x in nbytes],
               bitstream_md5=hashlib.md5(bs.tobytes()).hexdigest(),
               recon_md5=[bytes(m).hex() for m in md5],
               recon_md5_hashlib=[hashlib.md5(r.tobytes()).hexdigest() for r in rec])
    with open(os.path.join(HERE, "intra_cfg1.json"), "w") as f:
        json.dump(doc, f, indent=1)
    # the luma recon of frame 0 at CU16: lets the GPU inverse path be checked on real encoder data
    print("intra_cfg1.json:", doc["recon_md5"][0], "...", doc["recon_md5"][-1])


def gen_intra_decide(ref):
    """The reference's own luma mode decisions, recorded from inside xEncEncode (oracle/ref_shim.cpp
    ref_intra_record: the luma entry of xSadN[] is wrapped): per CU the padded reference samples, their smoothed
    copy, the most probable modes, the SAD and an FNV-1a hash of the prediction of each of the 35 modes and the
    mode the encoder kept.  Two encodes: BASELINE.json configs[0] (416x240, CU 16, QP 32; frames 0 and 5) and a
    CU-32 probe (416x224, QP 27, frame 1).  The source frames are synth.make_intra_probe's and are not stored."""
    out, meta = {}, []
    for (w, h, cu, qp, frames) in ((416, 240, 16, 32, (0, 5)), (416, 224, 32, 27, (1,))):
        yuv = synth.make_intra_probe(w, h, max(frames) + 1)
        for fr in frames:
            rec = ref.intra_record(yuv, w, h, cu, qp, fr)
            assert len(rec) == (w // cu) * (h // cu)
            k = f"w{w}h{h}cu{cu}qp{qp}f{fr}"
            nref = 4 * cu + 1
            out[k + "_xy"] = np.stack([rec["x"], rec["y"]], axis=1).astype(np.int32)
            out[k + "_ref0"] = np.ascontiguousarray(rec["ref0"][:, :nref])
            out[k + "_ref1"] = np.ascontiguousarray(rec["ref1"][:, :nref])
            out[k + "_mpm"] = np.ascontiguousarray(rec["mpm"])
            out[k + "_best"] = np.ascontiguousarray(rec["best_mode"])
            out[k + "_sad"] = np.ascontiguousarray(rec["sad"]).astype(np.uint32)
            out[k + "_fnv"] = np.ascontiguousarray(rec["pred_fnv"]).astype(np.uint32)
            meta.append(dict(key=k, width=w, height=h, cu=cu, qp=qp, frame=fr, ncus=int(len(rec)),
                             input_md5=hashlib.md5(yuv[fr].tobytes()).hexdigest()))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "intra_decide.npz"), **out)
    print("intra_decide.npz:", [m["key"] for m in meta])


def random_syntax(rng, w, h, cu, nf, style):
    """Levels and modes no encoder run produced -- every luma mode, every chroma candidate, empty blocks, lone
    levels anywhere in a block, dense blocks, magnitudes up to the Int16 range (escape codes)."""
    ncu = (w // cu) * (h // cu)
    lv = np.zeros((nf, w * h * 3 // 2), np.int16)
    my = rng.integers(0, 35, (nf, ncu)).astype(np.uint8)
    mc = np.zeros((nf, ncu), np.uint8)
    for f in range(nf):
        for i in range(ncu):
            cand = [0, 26, 10, 1, int(my[f, i])]          # src/encode.cpp:781-793 without LM
            for k in range(4):
                if cand[k] == my[f, i]:
                    cand[k] = 34
                    break
            mc[f, i] = cand[int(rng.integers(0, 5))]
        for pw, ph, n, off in ((w, h, cu, 0), (w // 2, h // 2, cu // 2, w * h), (w // 2, h // 2, cu // 2, w * h * 5 // 4)):
            pl = np.zeros((ph, pw), np.int64)
            for by in range(0, ph, n):
                for bx in range(0, pw, n):
                    r = rng.random()
                    if r < 0.15:
                        continue
                    if r < 0.3:
                        pl[by + rng.integers(0, n), bx + rng.integers(0, n)] = rng.integers(1, 4) * rng.choice([-1, 1])
                        continue
                    if style == "dense":
                        mask = np.ones((n, n), bool)
                    else:
                        fall = np.exp(-np.add.outer(np.arange(n), np.arange(n)) / (1 + rng.random() * n))
                        mask = rng.random((n, n)) < rng.random() * (1.0 if rng.random() < 0.2 else fall)
                    big = rng.integers(-32767, 32768, (n, n)) if style == "tail" else rng.integers(-300, 301, (n, n))
                    v = np.where(rng.random((n, n)) < (0.9 if style == "tail" else 0.6), rng.integers(-2, 3, (n, n)), big)
                    if style == "ones":
                        v = rng.choice([-1, 1], (n, n))
                    pl[by:by + n, bx:bx + n] = np.where(mask, v, 0)
            lv[f, off:off + pw * ph] = pl.ravel()
    return lv, my, mc


def gen_intra_stream(ref):
    """Pins x265amd_intra_write_picture: (1) the reference's own writer (headers, xEncIntraLoadRef's most probable
    modes, xWriteCU, CABAC: oracle/ref_shim.cpp ref_write_intra_frames) on random syntax; (2) a CU-32 encode of the
    reference itself (416x224, QP 27, 2 frames of synth.make_intra_probe), as intra_cfg1.json is for CU 16."""
    rng = np.random.default_rng(20240607)
    out, meta = {}, []
    for (w, h, cu, qp, nf, style) in ((64, 48, 16, 32, 3, "sparse"), (96, 64, 32, 20, 2, "sparse"), (48, 32, 16, 0, 2, "dense"),
                                      (32, 32, 16, 22, 2, "tail"), (64, 64, 32, 51, 1, "ones"), (64, 32, 32, 37, 2, "tail")):
        lv, my, mc = random_syntax(rng, w, h, cu, nf, style)
        zero = np.zeros((nf, w * h * 3 // 2), np.uint8)
        nbytes, bs = ref.write_intra_frames(zero, w, h, cu, qp, zero, lv, my, mc)
        k = f"w{w}h{h}cu{cu}qp{qp}{style}"
        out[k + "_levels"], out[k + "_mode_y"], out[k + "_mode_c"], out[k + "_bits"] = lv, my, mc, np.asarray(bs, np.uint8)
        meta.append(dict(key=k, width=w, height=h, cu=cu, qp=qp, frames=nf, bytes_per_frame=[int(x) for x in nbytes]))
    w, h, cu, qp, nf = 416, 224, 32, 27, 2
    yuv = synth.make_intra_probe(w, h, nf)
    rec, nbytes, md5, bs = ref.encode_intra(yuv, w, h, cu, qp)
    enc = dict(width=w, height=h, cu=cu, qp=qp, frames=nf, input_md5=hashlib.md5(yuv.tobytes()).hexdigest(),
               bytes_per_frame=[int(x) for x in nbytes], bitstream_md5=hashlib.md5(bs.tobytes()).hexdigest(),
               recon_md5=[hashlib.md5(r.tobytes()).hexdigest() for r in rec])
    out["meta"] = np.frombuffer(json.dumps(dict(random=meta, encode_cu32=enc)).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "intra_stream.npz"), **out)
    print("intra_stream.npz:", [m["key"] for m in meta], enc["bitstream_md5"])


def main():
    pyoracle.build()
    ref = pyoracle.Reference()
    gen_primitives(ref)
    gen_me(ref)
    gen_tq_frame(ref)
    gen_intra_cfg1(ref)
    gen_intra_decide(ref)
    gen_intra_stream(ref)


if __name__ == "__main__":
    main()
