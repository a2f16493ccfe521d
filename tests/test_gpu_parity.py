"""GPU parity: the HIP kernels, called through the C-ABI, against the golden
vectors from the compiled reference and against the CPU oracle on seeded
inputs.  Bit-exact everywhere (integer / byte / index work).

Run on the MI355X box:  python -u -m pytest tests -m gpu -x -v
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

MODE_INVALID = 255


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- SAD -----
def test_sad_golden(gpu):
    z, meta = load_golden("primitives.npz")
    for m in [m for m in meta if m["kind"] == "sad"]:
        k = m["id"]
        off = torch.zeros(1, dtype=torch.int32, device="cuda")
        got = gpu.sad_blocks(m["w"], m["n"], dev(z[f"sad{k}_src"]), m["ss"], dev(z[f"sad{k}_ref"]), m["sr"], off, off)
        assert int(host(got)[0]) == int(z[f"sad{k}_val"]), m


@pytest.mark.parametrize("w", [4, 8, 16, 32, 64])
def test_sad_batch_vs_oracle(gpu, oracle, w):
    """Many blocks at arbitrary (unaligned) offsets of two planes with different strides."""
    rng = np.random.default_rng(w)
    ss, sr, rows = 200, 312, 160
    src = rng.integers(0, 256, rows * ss, dtype=np.uint8)
    ref = rng.integers(0, 256, rows * sr, dtype=np.uint8)
    n = int(rng.integers(1, 65))
    nb = 300
    so = (rng.integers(0, rows - 64, nb) * ss + rng.integers(0, ss - 64, nb)).astype(np.int32)
    ro = (rng.integers(0, rows - 64, nb) * sr + rng.integers(0, sr - 64, nb)).astype(np.int32)
    got = host(gpu.sad_blocks(w, n, dev(src), ss, dev(ref), sr, dev(so), dev(ro)))
    want = [oracle.sad(w, n, src[so[b]:], ss, ref[ro[b]:], sr) for b in range(nb)]
    np.testing.assert_array_equal(got.astype(np.uint32), np.array(want, np.uint32))


# ---------------------------------------------------------------- TQ ------
def _block_view(a, n, dtype):
    """golden block buffer (stride 32) -> device [n][n] view with row stride 32"""
    t = dev(a.view(dtype).reshape(n, 32))
    return t[:, :n]


def test_tq_chain_golden(gpu):
    """Stage by stage like the reference's CHECK_TV blocks: transform, quant, dequant, recon."""
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "tq"]
    for m in cases:
        k, n = m["id"], m["n"]
        v = lambda a: a.reshape(n, 32)[:, :n]
        src = _block_view(z[f"tq{k}_src"], n, np.uint8)
        pred = _block_view(z[f"tq{k}_pred"], n, np.uint8)
        modes = None if m["mode"] == MODE_INVALID else torch.full((1, 1), m["mode"], dtype=torch.uint8, device="cuda")
        level, sums, coef = gpu.tq_forward(src, pred, n, m["qp"], m["slice"], modes=modes, want_coef=True)
        np.testing.assert_array_equal(host(coef)[0], v(z[f"tq{k}_coef"]), f"coef {m}")
        np.testing.assert_array_equal(host(level)[0], v(z[f"tq{k}_level"]), f"level {m}")
        assert int(host(sums)[0, 0]) == int(z[f"tq{k}_sum"]), m
        deq = gpu.dequant_plane(level[0], n, m["qp"])
        np.testing.assert_array_equal(host(deq), v(z[f"tq{k}_deq"]), f"deq {m}")
        qlevel, qsum = gpu.quant_plane(coef[0], n, m["qp"], m["slice"])
        np.testing.assert_array_equal(host(qlevel), v(z[f"tq{k}_level"]), f"quant_plane {m}")
        assert int(host(qsum)[0]) == int(z[f"tq{k}_sum"]), m
        rec = gpu.tq_inverse(level, pred, n, m["qp"], m["slice"], modes=modes)
        np.testing.assert_array_equal(host(rec)[0], v(z[f"tq{k}_rec"]), f"rec {m}")
        l2, s2, r2 = gpu.tq_roundtrip(src, pred, n, m["qp"], m["slice"], modes=modes)
        np.testing.assert_array_equal(host(l2)[0], v(z[f"tq{k}_level"]), f"rt level {m}")
        np.testing.assert_array_equal(host(r2)[0], v(z[f"tq{k}_rec"]), f"rt rec {m}")
        assert int(host(s2)[0, 0]) == int(z[f"tq{k}_sum"]), m


def test_tq_frame_golden(gpu):
    z, meta = load_golden("tq_frame.npz")
    cur, refp, mv = z["cur"], z["refp"], z["mv"]
    h, w = cur.shape
    dcur, drefp = dev(cur), dev(refp)
    dmv = dev(mv.view(np.int32).reshape(1, -1, 2))
    for m in meta:
        k, n = m["id"], m["n"]
        kw = dict(pred_origin=(32, 32))
        if m["mv"]:
            kw.update(mv=dmv, mv_ctu=64)
        level, sums = gpu.tq_forward(dcur, drefp, n, m["qp"], m["slice"], **kw)
        rec = gpu.tq_inverse(level, drefp, n, m["qp"], m["slice"], **kw)
        l2, s2, r2 = gpu.tq_roundtrip(dcur, drefp, n, m["qp"], m["slice"], **kw)
        for got, name in ((level, "level"), (l2, "level")):
            np.testing.assert_array_equal(host(got)[0], z[f"f{k}_level"], f"{name} {m}")
        for got in (sums, s2):
            np.testing.assert_array_equal(host(got)[0].astype(np.uint32), z[f"f{k}_sums"], str(m))
        for got in (rec, r2):
            np.testing.assert_array_equal(host(got)[0], z[f"f{k}_rec"], f"rec {m}")


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_tq_ragged_planes_vs_oracle(gpu, oracle, n):
    """Plane sizes that are multiples of the TU but not of the 32x32 work tile, batched
    over frames, with per-TU intra modes (DST only matters at 4x4)."""
    from x265_amd import synth
    rng = np.random.default_rng(n)
    for (w, h) in ((n, n), (n * 3, n * 5), (32 + n, 64 + n), (96, 40 if n <= 8 else 64)):
        if w % n or h % n:
            continue
        f = 3
        seq = synth.make_sequence(max(w, 64), max(h, 64), f + 1, search_range=4, seed=w * 131 + h)
        src = np.ascontiguousarray(seq[1:, :h, :w])
        pred = np.ascontiguousarray(seq[:-1, :h, :w])
        ntu = (h // n) * (w // n)
        modes = rng.integers(0, 36, (f, ntu)).astype(np.uint8)
        modes[modes == 35] = MODE_INVALID
        qp, st = int(rng.integers(0, 52)), int(rng.integers(0, 2))
        level, sums, rec = gpu.tq_roundtrip(dev(src), dev(pred), n, qp, st, modes=dev(modes))
        rec_i = gpu.tq_inverse(level, dev(pred), n, qp, st, modes=dev(modes))
        for i in range(f):
            ol, os_ = oracle.tq_forward(src[i], pred[i], n, qp, st, modes=modes[i])
            orc = oracle.tq_inverse(ol, pred[i], n, qp, st, modes=modes[i])
            np.testing.assert_array_equal(host(level)[i], ol, f"level {w}x{h} f{i}")
            np.testing.assert_array_equal(host(sums)[i].astype(np.uint32), os_, f"sum {w}x{h} f{i}")
            np.testing.assert_array_equal(host(rec)[i], orc, f"rec {w}x{h} f{i}")
            np.testing.assert_array_equal(host(rec_i)[i], orc, f"inv rec {w}x{h} f{i}")


def test_tq_inverse_illegal_levels_wrap_like_reference(gpu, oracle):
    """Levels no encoder would produce: the inverse must wrap exactly as the reference's
    store-then-clip does, not saturate."""
    rng = np.random.default_rng(5)
    for n in (4, 8, 16, 32):
        h, w = 64, 64
        level = rng.integers(-32768, 32768, (h, w)).astype(np.int16)
        pred = rng.integers(0, 256, (h, w)).astype(np.uint8)
        for qp in (0, 23, 51):
            got = host(gpu.tq_inverse(dev(level), dev(pred), n, qp))[0]
            np.testing.assert_array_equal(got, oracle.tq_inverse(level, pred, n, qp), f"n={n} qp={qp}")


def test_tq_1080p_all_sizes_vs_oracle(gpu, oracle):
    """BASELINE.json configs[2] at full size: 1920x1088 luma, pred = co-located block of the
    previous frame, QP 32, SLICE_I rounding -- every TU size, against the CPU oracle."""
    import hashlib
    from x265_amd import synth
    seq = synth.make_sequence(1920, 1088, 2, search_range=8, seed=12345)
    cur, prev = dev(seq[1]), dev(seq[0])
    for n in (32, 16, 8, 4):
        level, sums, rec = gpu.tq_roundtrip(cur, prev, n, 32, 0)
        ol, os_ = oracle.tq_forward(seq[1], seq[0], n, 32, 0, nthreads=8)
        orc = oracle.tq_inverse(ol, seq[0], n, 32, 0, nthreads=8)
        np.testing.assert_array_equal(host(level)[0], ol)
        np.testing.assert_array_equal(host(sums)[0].astype(np.uint32), os_)
        assert hashlib.md5(host(rec)[0].tobytes()).hexdigest() == hashlib.md5(orc.tobytes()).hexdigest()


def test_tq_argument_errors(gpu):
    a = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.X265AmdError):
        gpu.tq_roundtrip(a, a, 64, 32)          # the reference asserts TU <= 32 (src/pixel.cpp:684)
    with pytest.raises(gpu.X265AmdError):
        gpu.tq_roundtrip(a, a, 8, 52)           # QP out of range
    with pytest.raises(gpu.X265AmdError):
        gpu.tq_roundtrip(a[:60], a[:60], 8, 30)  # height not a multiple of the TU (src/params.cpp:74)


# ---------------------------------------------------------------- ME ------
@pytest.mark.parametrize("method", ["brute", "sea"])
def test_me_golden(gpu, method):
    z, meta = load_golden("me_small.npz")
    for m in meta:
        k = m["id"]
        if method == "sea" and m["ctu"] not in (64, 32):
            continue
        got = gpu.mv_to_numpy(gpu.me_fullsearch(dev(z[f"me{k}_cur"]), dev(z[f"me{k}_ref"]), m["ctu"], m["range"],
                                                method=method))
        want = z[f"me{k}_mv"]
        for f in ("mvx", "mvy", "sad"):
            np.testing.assert_array_equal(got[0, 0][f], want[f], f"{f} {m}")


@pytest.mark.parametrize("method", ["brute", "sea"])
@pytest.mark.parametrize("ctu,rng_,w,h", [(64, 64, 256, 192), (64, 33, 320, 128), (32, 24, 160, 96),
                                          (16, 16, 96, 64), (64, 128, 192, 192), (64, 1, 128, 64),
                                          (64, 0, 128, 64), (32, 64, 256, 160), (32, 128, 192, 160), (32, 1, 96, 32),
                                          (32, 0, 64, 64), (32, 37, 416, 96)])
def test_me_vs_oracle(gpu, oracle, ctu, rng_, w, h, method):
    from x265_amd import synth
    if method == "sea" and ctu not in (64, 32):
        pytest.skip("successive elimination is built for 64x64 and 32x32 blocks")
    seq = synth.make_sequence(w, h, 3, search_range=min(rng_, 48), seed=ctu * 1000 + rng_)
    cur = np.ascontiguousarray(seq[1:])
    ref = np.ascontiguousarray(seq[:-1])
    got = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, rng_, method=method))
    for f in range(2):
        want = oracle.me_fullsearch(cur[f], ref[f], ctu, rng_, nthreads=8)
        np.testing.assert_array_equal(got[f, 0], want, f"frame {f}")


@pytest.mark.parametrize("method", ["brute", "sea"])
def test_me_ties_pick_first_in_raster_order(gpu, oracle, method):
    """Periodic content => many exact ties; the winner must be the first candidate in
    (dy outer, dx inner) order -- the reference's strict `<` (src/encode.cpp:762)."""
    w, h = 256, 128
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((xx % 8) * 16 + (yy % 4) * 3) & 255).astype(np.uint8)
    got = gpu.mv_to_numpy(gpu.me_fullsearch(dev(img), dev(img.copy()), 64, 20, method=method))[0, 0]
    want = oracle.me_fullsearch(img, img.copy(), 64, 20, nthreads=8)
    np.testing.assert_array_equal(got, want)
    assert (want["sad"] == 0).all() and (want["mvy"] <= 0).all()


@pytest.mark.parametrize("method", ["brute", "sea"])
def test_me_multi_reference(gpu, oracle, method):
    """BASELINE.json configs[4] shape at small size: 4 reference frames per current frame."""
    from x265_amd import synth
    seq = synth.make_sequence(192, 128, 6, search_range=16, seed=31)
    f, nref = 2, 4
    cur = np.ascontiguousarray(seq[4:6])
    refs = np.stack([np.stack([seq[4 + i - 1 - r] for r in range(nref)]) for i in range(f)])
    got = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(refs), 64, 24, method=method))
    for i in range(f):
        for r in range(nref):
            want = oracle.me_fullsearch(cur[i], np.ascontiguousarray(refs[i, r]), 64, 24, nthreads=8)
            np.testing.assert_array_equal(got[i, r], want, f"frame {i} ref {r}")


def _me_fullsize_properties(gpu, oracle, method, W, H, R, first, step):
    """(a) exact comparison with the CPU oracle on CTUs first, first+step, ..; (b) for ALL CTUs the
    returned cost is the true SAD at the returned vector (recomputed by the independent batched-SAD
    kernel), the vector is legal, and no worse than the zero vector and 8 random legal candidates;
    (c) the two entry points agree on every CTU."""
    from x265_amd import synth
    seq = synth.make_sequence(W, H, 2, search_range=R, seed=12345)
    cur, ref = dev(seq[1]), dev(seq[0])
    mv = gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 64, R, method=method))[0, 0]
    other = gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 64, R, method="brute" if method == "sea" else "sea"))[0, 0]
    np.testing.assert_array_equal(mv, other)
    ncx = W // 64
    nctu = mv.size
    want = oracle.me_fullsearch(seq[1], seq[0], 64, R, first=first, step=step, count=-1, nthreads=16)
    idx = np.arange(first, nctu, step)
    np.testing.assert_array_equal(mv[idx], want[idx])
    cx = (np.arange(nctu) % ncx) * 64
    cy = (np.arange(nctu) // ncx) * 64
    assert ((cx + mv["mvx"] >= 0) & (cx + mv["mvx"] + 64 <= W) & (cy + mv["mvy"] >= 0) & (cy + mv["mvy"] + 64 <= H)).all()
    assert (np.abs(mv["mvx"]) <= R).all() and (np.abs(mv["mvy"]) <= R).all()
    so = dev((cy * W + cx).astype(np.int32))
    ro = dev(((cy + mv["mvy"]) * W + cx + mv["mvx"]).astype(np.int32))
    sad_at_mv = host(gpu.sad_blocks(64, 64, cur.view(-1), W, ref.view(-1), W, so, ro)).astype(np.uint32)
    np.testing.assert_array_equal(sad_at_mv, mv["sad"])
    rng = np.random.default_rng(0)
    for t in range(9):
        dx = np.zeros(nctu, np.int64) if t == 0 else rng.integers(-R, R + 1, nctu)
        dy = np.zeros(nctu, np.int64) if t == 0 else rng.integers(-R, R + 1, nctu)
        dx = np.clip(cx + dx, 0, W - 64) - cx
        dy = np.clip(cy + dy, 0, H - 64) - cy
        ro = dev(((cy + dy) * W + cx + dx).astype(np.int32))
        s = host(gpu.sad_blocks(64, 64, cur.view(-1), W, ref.view(-1), W, so, ro)).astype(np.uint32)
        assert (mv["sad"] <= s).all()


@pytest.mark.parametrize("method", ["brute", "sea"])
def test_me_1080p_properties(gpu, oracle, method):
    """BASELINE.json configs[1] at full size (1920x1088, CTU 64, R 64); the CPU oracle needs ~0.1 s per
    CTU, hence every 17th CTU."""
    _me_fullsize_properties(gpu, oracle, method, 1920, 1088, 64, 3, 17)


def test_me_4k_properties(gpu, oracle):
    """BASELINE.json configs[3] at full size (3840x2176, R 128): the +-128 window leaves room for one
    workgroup per CU only, which is the 16-wave shape of the exact kernel.  Oracle on every 97th CTU."""
    _me_fullsize_properties(gpu, oracle, "sea", 3840, 2176, 128, 5, 97)


def test_me_sea_equals_exhaustive_on_hostile_content(gpu):
    """Successive elimination must return exactly what the exhaustive kernel returns whatever
    the content: white noise (bounds useless), flat (everything ties), a 1080p frame pair with
    unrelated content (no good match anywhere), and real motion."""
    from x265_amd import synth
    rng = np.random.default_rng(11)
    cases = []
    cases.append((rng.integers(0, 256, (192, 256), dtype=np.uint8), rng.integers(0, 256, (192, 256), dtype=np.uint8), 40))
    flat = np.full((128, 192), 77, np.uint8)
    cases.append((flat, flat.copy(), 33))
    a = synth.make_sequence(1920, 1088, 1, seed=1)[0]
    b = synth.make_sequence(1920, 1088, 1, seed=2)[0]
    cases.append((a, np.ascontiguousarray(b[::-1]), 64))
    seq = synth.make_sequence(640, 384, 2, search_range=64, seed=5)
    cases.append((seq[1], seq[0], 64))
    # +-128: the 16-wave shape, whose second pass drops groups on the SAD of the block's upper half -- noise (every
    # group outlives it: the kept list is drained again and again), flat (all tie), and real motion
    cases.append((rng.integers(0, 256, (320, 384), dtype=np.uint8), rng.integers(0, 256, (320, 384), dtype=np.uint8), 128))
    flat2 = np.full((256, 320), 200, np.uint8)
    cases.append((flat2, flat2.copy(), 128))
    seq = synth.make_sequence(512, 384, 2, search_range=128, seed=9)
    cases.append((seq[1], seq[0], 128))
    for cur, ref, r in cases:
        for ctu in (64, 32):
            x = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, r, method="brute"))
            y = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, r, method="sea"))
            np.testing.assert_array_equal(x, y, f"ctu {ctu} range {r}")


@pytest.mark.parametrize("noise_rows", [1, 3, 5, 7, 8])
def test_me_sea_partly_unprunable_pictures(gpu, noise_rows):
    """Pictures of which a part is white noise (those blocks go to the every-candidate list of the elimination search) and
    the rest is content that moves.  A few listed blocks are cut into slices (1 and 3 rows of 8: eight and two slices a
    block); when more than half of the blocks are listed (5, 7, 8 rows) they are taken block by block in the XCD-aware
    order and the unlisted ones skipped.  Every share must give what the every-candidate kernel gives."""
    from x265_amd import synth
    rng = np.random.default_rng(40 + noise_rows)
    h, w, r = 512, 576, 32
    seq = synth.make_sequence(w, h, 3, search_range=r, seed=70 + noise_rows)
    cur = np.stack([seq[1], seq[2]]).copy()
    ref = np.stack([seq[0], seq[1]]).copy()
    for f in range(2):
        cur[f, :64 * noise_rows] = rng.integers(0, 256, (64 * noise_rows, w), dtype=np.uint8)
        ref[f, :64 * noise_rows] = rng.integers(0, 256, (64 * noise_rows, w), dtype=np.uint8)
    for ctu in (64, 32):
        x = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, r, method="brute"))
        y = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, r, method="sea"))
        np.testing.assert_array_equal(x, y, f"ctu {ctu}")


def test_me_argument_errors(gpu):
    a = torch.zeros((128, 128), dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(a, a, 48, 16)         # CTU size not supported
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(a[:100], a[:100], 64, 16)  # height not a multiple of the CTU
    big = torch.zeros((1024, 1024), dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(big, big, 64, 300)    # (64 + 600)^2 window cannot fit the 160 KiB LDS


def test_me_then_tq_pipeline_vs_oracle(gpu, oracle):
    """The hot path end to end on a small picture: search -> integer-pel compensated
    prediction -> forward TQ -> reconstruction; MVs, costs, levels and recon MD5 equal the oracle's."""
    import hashlib
    from x265_amd import synth
    seq = synth.make_sequence(256, 192, 2, search_range=24, seed=99)
    cur, ref = dev(seq[1]), dev(seq[0])
    mv = gpu.me_fullsearch(cur, ref, 64, 24)
    level, sums, rec = gpu.tq_roundtrip(cur, ref, 32, 32, 1, mv=mv.view(1, -1, 2), mv_ctu=64)
    omv = oracle.me_fullsearch(seq[1], seq[0], 64, 24, nthreads=8)
    np.testing.assert_array_equal(gpu.mv_to_numpy(mv)[0, 0], omv)
    ol, os_ = oracle.tq_forward(seq[1], seq[0], 32, 32, 1, mv=omv, mv_ctu=64)
    orc = oracle.tq_inverse(ol, seq[0], 32, 32, 1, mv=omv, mv_ctu=64)
    np.testing.assert_array_equal(host(level)[0], ol)
    np.testing.assert_array_equal(host(sums)[0].astype(np.uint32), os_)
    assert hashlib.md5(host(rec)[0].tobytes()).hexdigest() == hashlib.md5(orc.tobytes()).hexdigest()


def test_closed_loop_gop_vs_oracle(gpu, oracle):
    """closed-loop P pictures with xEncEncode's reference rotation
    (src/encode.cpp:619-621,965-970), two GOP segments batched.  The CPU chain is the oracle's search
    and TQ applied picture by picture to its OWN reconstruction; vectors, levels and the MD5 of every
    reconstructed Y+U+V picture (what src/encode.cpp:945-950 hashes) must agree."""
    import hashlib
    from x265_amd import synth, ClosedLoopGop
    W, H, R, QP, T = 256, 128, 16, 30, 4
    segs = []
    for seed in (5, 6):
        y, u, v = synth.make_sequence(W, H, T, search_range=R, seed=seed, chroma=True)
        segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
    yuv = np.stack(segs)                                   # [S][T][W*H*3/2]
    S, ysz, csz = 2, W * H, (W // 2) * (H // 2)
    enc = ClosedLoopGop(W, H, S, search_range=R, qp=QP)
    enc.set_reference(dev(yuv[:, 0]))                      # the I picture's reconstruction: here the source itself
    qpc = oracle.chroma_qp(QP)
    ref_cpu = [yuv[s, 0].copy() for s in range(S)]
    for t in range(1, T):
        rec_gpu = host(enc.encode_p(dev(yuv[:, t])))
        gmv = gpu.mv_to_numpy(enc.mv)[:, 0]
        glevel = host(enc.level)
        for s in range(S):
            cur, ref = yuv[s, t], ref_cpu[s]
            cy, ry = cur[:ysz].reshape(H, W), ref[:ysz].reshape(H, W)
            omv = oracle.me_fullsearch(cy, ry, 64, R, nthreads=8)
            np.testing.assert_array_equal(gmv[s], omv)
            ly, _ = oracle.tq_forward(cy, ry, 32, QP, 1, mv=omv, mv_ctu=64)
            rec = [oracle.tq_inverse(ly, ry, 32, QP, 1, mv=omv, mv_ctu=64).reshape(-1)]
            lev = [ly.reshape(-1)]
            cmv = omv.copy()
            cmv["mvx"] >>= 1
            cmv["mvy"] >>= 1
            for k in range(2):
                a, b = ysz + k * csz, ysz + (k + 1) * csz
                cc, rc = cur[a:b].reshape(H // 2, W // 2), ref[a:b].reshape(H // 2, W // 2)
                lc, _ = oracle.tq_forward(cc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32)
                rec.append(oracle.tq_inverse(lc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32).reshape(-1))
                lev.append(lc.reshape(-1))
            ref_cpu[s] = np.concatenate(rec)
            np.testing.assert_array_equal(glevel[s], np.concatenate(lev))
            assert hashlib.md5(rec_gpu[s].tobytes()).hexdigest() == hashlib.md5(ref_cpu[s].tobytes()).hexdigest(), (s, t)
    assert enc.poc == T - 1


def test_me_sea_strides_batches_and_workspace(gpu, oracle):
    """The successive-elimination entry point on what the pipeline never exercises: a row stride wider
    than the picture, a batch whose frames are not back to back, a caller workspace that is reused, and
    the refusals (workspace too small or misaligned, CTU other than 64)."""
    from x265_amd import synth, api
    W, H, R, F = 192, 128, 20, 3
    seq = synth.make_sequence(W, H, F + 1, search_range=R, seed=41)
    big = torch.zeros((F + 1, H + 8, W + 64), dtype=torch.uint8, device="cuda")   # padded rows and frames
    big[:, :H, :W] = dev(seq)
    frames = big[:, :H, :W]                                                       # stride W + 64, frame stride (H + 8) * (W + 64)
    ws = torch.empty(api.me_workspace_bytes(W, H, R, F, 1) + 64, dtype=torch.uint8, device="cuda")
    for _ in range(2):                                                            # the second call reuses the workspace as it was left
        mv = gpu.me_fullsearch(frames[1:], frames[:-1], 64, R, method="sea", workspace=ws)
        got = gpu.mv_to_numpy(mv)
        for f in range(F):
            np.testing.assert_array_equal(got[f, 0], oracle.me_fullsearch(seq[f + 1], seq[f], 64, R, nthreads=8))
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(frames[1:], frames[:-1], 64, R, method="sea", workspace=ws[:1024])   # too small
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(frames[1:], frames[:-1], 64, R, method="sea", workspace=ws[4:])      # not 16-byte aligned
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(frames[1:], frames[:-1], 16, R, method="sea", workspace=ws)          # SEA: blocks of 64 or 32
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(frames[1:], frames[:-1], 32, R, method="sea", workspace=ws)          # 32x32 blocks need a larger workspace


def test_tq_mv_shift_chroma_vs_oracle(gpu, oracle):
    """mv_shift of the TQ descriptor: chroma planes predicted with the luma vectors >> 1 (arithmetic shift,
    negative odd vectors included) equal the oracle fed with the halved vectors."""
    rng = np.random.default_rng(8)
    H, W = 64, 96                                        # chroma plane of a 192x128 picture: 3 x 2 blocks of 32
    src = rng.integers(0, 256, (H, W), dtype=np.uint8)
    ref = rng.integers(0, 256, (H, W), dtype=np.uint8)
    mv = np.zeros(6, dtype=gpu.MV_DTYPE)
    mv["mvx"] = [-7, 5, -64, 63, 1, -1]                  # luma units; block x0 = 0 / 32 / 64 keeps them inside
    mv["mvy"] = [3, -5, 9, -63, -1, 64]
    # keep every displaced 32x32 block inside the plane
    for i in range(6):
        bx, by = (i % 3) * 32, (i // 3) * 32
        mv["mvx"][i] = np.clip(mv["mvx"][i] >> 1, -bx, W - 32 - bx) * 2 + (mv["mvx"][i] & 1)
        mv["mvy"][i] = np.clip(mv["mvy"][i] >> 1, -by, H - 32 - by) * 2 + (mv["mvy"][i] & 1)
    half = mv.copy()
    half["mvx"] >>= 1
    half["mvy"] >>= 1
    dmv = torch.from_numpy(mv.view(np.int32).reshape(1, 6, 2).copy()).cuda()
    level, sums, rec = gpu.tq_roundtrip(dev(src), dev(ref), 16, 29, 1, mv=dmv, mv_ctu=32, mv_shift=1)
    ol, os_ = oracle.tq_forward(src, ref, 16, 29, 1, mv=half, mv_ctu=32)
    orec = oracle.tq_inverse(ol, ref, 16, 29, 1, mv=half, mv_ctu=32)
    np.testing.assert_array_equal(host(level)[0], ol)
    np.testing.assert_array_equal(host(sums)[0].astype(np.uint32), os_)
    np.testing.assert_array_equal(host(rec)[0], orec)


def test_tq_check_mv_rejects_vectors_that_leave_the_plane(gpu, oracle):
    """X265AMD_TQ_CHECK_MV (x265amd_tq_desc.flags): a vector that takes its block outside the prediction plane is
    refused with X265AMD_ERR_ARG before anything is transformed (the outputs keep their contents); legal vectors go
    through and give the oracle's planes; unknown flag bits are refused."""
    rng = np.random.default_rng(81)
    H, W = 128, 192
    src = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    ref = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    mv = np.zeros((2, 6), dtype=gpu.MV_DTYPE)
    for f in range(2):
        for i in range(6):
            bx, by = (i % 3) * 64, (i // 3) * 64
            mv["mvx"][f, i] = rng.integers(-bx, W - 64 - bx + 1)
            mv["mvy"][f, i] = rng.integers(-by, H - 64 - by + 1)
    dmv = torch.from_numpy(mv.view(np.int32).reshape(2, 6, 2).copy()).cuda()
    level, sums, rec = gpu.tq_roundtrip(dev(src), dev(ref), 32, 30, 1, mv=dmv, mv_ctu=64, check_mv=True)
    for f in range(2):
        ol, os_ = oracle.tq_forward(src[f], ref[f], 32, 30, 1, mv=mv[f], mv_ctu=64)
        np.testing.assert_array_equal(host(level)[f], ol)
        np.testing.assert_array_equal(host(rec)[f], oracle.tq_inverse(ol, ref[f], 32, 30, 1, mv=mv[f], mv_ctu=64))
    for (f, i, field, val) in ((1, 5, "mvx", 1), (0, 0, "mvy", -1), (1, 2, "mvx", 32767), (0, 3, "mvy", 65)):
        bad = mv.copy()
        bad[field][f, i] = val if abs(val) > 1 else (W - 64 - (i % 3) * 64 + 1 if field == "mvx" else -((i // 3) * 64) - 1)
        dbad = torch.from_numpy(bad.view(np.int32).reshape(2, 6, 2).copy()).cuda()
        out = (torch.full_like(level, 77), torch.full_like(sums, 7), torch.full_like(rec, 9))
        with pytest.raises(gpu.X265AmdError) as e:
            gpu.tq_roundtrip(dev(src), dev(ref), 32, 30, 1, mv=dbad, mv_ctu=64, check_mv=True, out=out)
        assert e.value.status == -1                       # X265AMD_ERR_ARG
        torch.cuda.synchronize()
        assert (out[0] == 77).all() and (out[2] == 9).all()   # nothing was launched
    d = gpu.api._tq_desc(32, 30, 1, src=dev(src), pred=dev(ref), mv=dmv, mv_ctu=64, level=level, abs_sum=sums, rec=rec)
    d.flags = 2
    assert gpu.api._lib().x265amd_tq_roundtrip(gpu.api.C.byref(d), None) == -1


@pytest.mark.parametrize("ctu,rng_,w,h", [(64, 40, 320, 192), (32, 33, 224, 160), (16, 21, 144, 112)])
def test_me_production_kernel_equals_basic_formulation(gpu, ctu, rng_, w, h):
    """The qsad kernel (all three block sizes) against the independent one-candidate-per-lane
    formulation (v_alignbyte + v_sad_u8), random content so that ties and edge clamps both occur."""
    rng = np.random.default_rng(ctu)
    cur = rng.integers(0, 4, (2, h, w), dtype=np.uint8) * 60
    ref = rng.integers(0, 4, (2, h, w), dtype=np.uint8) * 60
    a = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, rng_, method="brute"))
    b = gpu.mv_to_numpy(gpu.me_fullsearch(dev(cur), dev(ref), ctu, rng_, method="basic"))
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("method", ["sea", "brute"])
def test_pipeline_multi_reference_negative_stride(gpu, oracle, method):
    """MeTqPipeline with 2 references: ref(f, r) = frame (f - 1 - r), handed to the C-ABI as a NEGATIVE
    byte stride between references (no copies).  Vectors per (frame, ref) against the oracle; the
    transform stage uses reference 0."""
    from x265_amd import synth, MeTqPipeline
    W, H, R, NR, F = 192, 128, 12, 2, 3
    seq = synth.make_sequence(W, H, F + NR, search_range=R, seed=77)
    pipe = MeTqPipeline(dev(seq), ctu=64, search_range=R, nrefs=NR, tu=32, qp=27, method=method)
    pipe.step()
    got = gpu.mv_to_numpy(pipe.mv)
    for f in range(F):
        for r in range(NR):
            want = oracle.me_fullsearch(seq[NR + f], seq[NR + f - 1 - r], 64, R, nthreads=8)
            np.testing.assert_array_equal(got[f, r], want, f"frame {f} ref {r}")
        lvl, sums = oracle.tq_forward(seq[NR + f], seq[NR + f - 1], 32, 27, 1, mv=got[f, 0], mv_ctu=64)
        rec = oracle.tq_inverse(lvl, seq[NR + f - 1], 32, 27, 1, mv=got[f, 0], mv_ctu=64)
        np.testing.assert_array_equal(host(pipe.level)[f], lvl)
        np.testing.assert_array_equal(host(pipe.rec)[f], rec)


# ---------------------------------------------------------------- PU tree --
def test_me_pu_tree_vs_oracle(gpu, oracle):
    """x265amd_me_fullsearch_pu: all square PU sizes in one pass == the oracle's search run per size.  Small
    picture: every CTU touches a border, so the per-PU legality (a PU may move where its CTU may not) is what
    is tested; range larger than the picture allows included."""
    from x265_amd import synth
    for (W, H, R, seed) in ((128, 64, 20, 3), (192, 128, 9, 4), (64, 64, 70, 5)):
        seq = synth.make_sequence(W, H, 2, search_range=min(R, 16), seed=seed)
        got = gpu.me_fullsearch_pu(dev(seq[1]), dev(seq[0]), R)
        for b in (64, 32, 16):
            want = oracle.me_fullsearch(seq[1], seq[0], b, R, nthreads=8)
            np.testing.assert_array_equal(gpu.mv_to_numpy(got[b])[0, 0], want, f"{W}x{H} R{R} PU {b}")


def test_me_pu_tree_equals_per_size_kernels_batched(gpu):
    """Frames x references batch, flat (all ties) and noise content: identical to x265amd_me_fullsearch per size."""
    from x265_amd import synth
    W, H, R, F = 320, 192, 24, 3
    seq = synth.make_sequence(W, H, F + 2, search_range=R, seed=8)
    seq[2, 64:128, 64:192] = 77                      # a flat patch: every candidate inside it ties
    rng = np.random.default_rng(1)
    seq[4, :, 192:] = rng.integers(0, 256, (H, W - 192), dtype=np.uint8)
    fr = dev(seq)
    cur = fr[2:]
    refs = torch.stack([fr[1:-1], fr[:-2]], dim=1)    # [F][2][H][W]
    got = gpu.me_fullsearch_pu(cur, refs, R, sizes=(32, 16, 64))
    for b in (64, 32, 16):
        want = gpu.me_fullsearch(cur, refs, b, R, method="brute")
        assert torch.equal(got[b], want), b
    only16 = gpu.me_fullsearch_pu(cur, refs, R, sizes=(16,))
    assert torch.equal(only16[16], got[16]) and list(only16) == [16]


def test_me_all_pu_shapes_vs_oracle(gpu, oracle):
    """x265amd_me_fullsearch_pu_all: the ten inter PU shapes (squares 64 .. 8, 2NxN and Nx2N at 64 / 32 / 16) in
    one pass == the oracle's exhaustive search of the w x h blocks tiling the picture (the reference's xSadN with
    h rows).  Small pictures: every CTU touches a border, so the per-PU legality is what is tested; a range larger
    than the picture allows, range 0 and a range that is not a multiple of the four-wide groups included; flat
    patches for the ties."""
    from x265_amd import synth
    for (W, H, R, seed) in ((128, 64, 20, 3), (192, 128, 9, 4), (64, 64, 70, 5), (128, 128, 0, 6), (192, 192, 33, 7)):
        seq = synth.make_sequence(W, H, 2, search_range=max(2, min(R, 16)), seed=seed)
        seq[1, 8:40, 16:80] = 90
        seq[0, 0:56, 8:100] = 90
        got = gpu.me_fullsearch_pu_all(dev(seq[1]), dev(seq[0]), R)
        assert list(got) == list(gpu.PU_SHAPES)
        for (bw, bh) in gpu.PU_SHAPES:
            want = oracle.me_fullsearch_wh(seq[1], seq[0], bw, bh, R, nthreads=16)
            np.testing.assert_array_equal(gpu.mv_to_numpy(got[(bw, bh)])[0, 0], want, f"{W}x{H} R{R} PU {bw}x{bh}")


def test_me_all_pu_shapes_interior_and_batched(gpu, oracle):
    """A picture with interior CTUs (the path without legality arithmetic) next to border ones, frames x
    references in one launch, range 64 (15-bit raster index) and 96 (17-bit: the 16x16 PUs leave the single-dword
    keys): squares identical to x265amd_me_fullsearch, rectangles against the oracle on a sample of blocks, and a
    subset of shapes gives the same planes as the full set."""
    from x265_amd import synth
    W, H, F = 448, 320, 2
    for R in (64, 96):
        seq = synth.make_sequence(W, H, F + 2, search_range=24, seed=60 + R)
        seq[2, 100:180, 120:260] = 33
        seq[1, 60:220, 80:300] = 33
        fr = dev(seq)
        cur = fr[2:]
        refs = torch.stack([fr[1:-1], fr[:-2]], dim=1)    # [F][2][H][W]
        got = gpu.me_fullsearch_pu_all(cur, refs, R)
        for b in (64, 32, 16):
            want = gpu.me_fullsearch(cur, refs, b, R, method="brute")
            assert torch.equal(got[(b, b)].reshape(want.shape), want), (R, b)
        for (bw, bh) in gpu.PU_SHAPES:
            if bw == bh and bw > 8:
                continue
            nb = (W // bw) * (H // bh)
            step = max(1, nb // 24)
            for fi, ri in ((0, 0), (1, 1)):
                c, r = seq[2 + fi], seq[2 + fi - 1 - ri]
                want = oracle.me_fullsearch_wh(c, r, bw, bh, R, first=fi, step=step, nthreads=16).reshape(-1)
                idx = np.arange(fi, nb, step)
                g = gpu.mv_to_numpy(got[(bw, bh)])[fi, ri].reshape(-1)
                np.testing.assert_array_equal(g[idx], want[idx], f"R{R} {bw}x{bh} f{fi} r{ri}")
        sub = gpu.me_fullsearch_pu_all(cur, refs, R, shapes=((8, 8), (64, 32)))
        assert list(sub) == [(8, 8), (64, 32)]
        assert torch.equal(sub[(8, 8)], got[(8, 8)]) and torch.equal(sub[(64, 32)], got[(64, 32)])


@pytest.mark.parametrize("R", [0, 1, 3, 7, 13, 33])
def test_me_odd_and_degenerate_ranges_vs_oracle(gpu, oracle, R):
    """Search ranges that are not multiples of the 4-wide candidate groups, down to a single candidate, on a
    picture where every CTU touches a border: elimination search, every-candidate kernel and PU tree."""
    from x265_amd import synth
    W, H = 192, 128
    seq = synth.make_sequence(W, H, 2, search_range=max(R, 2), seed=50 + R)
    want = oracle.me_fullsearch(seq[1], seq[0], 64, R, nthreads=8)
    cur, ref = dev(seq[1]), dev(seq[0])
    for method in ("sea", "brute"):
        got = gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 64, R, method=method))[0, 0]
        np.testing.assert_array_equal(got, want, f"R {R} {method}")
    tree = gpu.me_fullsearch_pu(cur, ref, R)
    np.testing.assert_array_equal(gpu.mv_to_numpy(tree[64])[0, 0], want)
    want32 = oracle.me_fullsearch(seq[1], seq[0], 32, R, nthreads=8)
    np.testing.assert_array_equal(gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 32, R, method="sea"))[0, 0], want32, f"R {R} sea 32")
    np.testing.assert_array_equal(gpu.mv_to_numpy(tree[16])[0, 0], oracle.me_fullsearch(seq[1], seq[0], 16, R, nthreads=8))


# ------------------------------------------------------ intra decision --
def test_intra_decide_luma_recorded_reference_decisions(gpu):
    """x265amd_intra_decide_luma against the reference encoder's own decisions (tests/golden/intra_decide.npz,
    recorded from inside xEncEncode: BASELINE.json configs[0] frames 0 and 5 at CU 16, a CU-32 probe): every CU of
    the frame in one launch -- the 35 SADs, the mode kept, its cost, and the winner's prediction by its hash."""
    from conftest import load_golden
    from oracle.pyoracle import fnv1a
    from x265_amd import synth
    z, meta = load_golden("intra_decide.npz")
    for m in meta:
        k, n, lam = m["key"], m["cu"], m["qp"]
        yuv = synth.make_intra_probe(m["width"], m["height"], m["frame"] + 1)
        Y = yuv[m["frame"], :m["width"] * m["height"]].reshape(m["height"], m["width"])
        best, sads, pred = gpu.intra_decide_luma(dev(Y), dev(z[k + "_xy"]), n, dev(z[k + "_ref0"]), dev(z[k + "_mpm"]), lam,
                                                 want_sads=True, want_pred=True)
        best, sads, pred = host(best), host(sads), host(pred)
        np.testing.assert_array_equal(sads.astype(np.uint32), z[k + "_sad"], k)
        np.testing.assert_array_equal(best[:, 0], z[k + "_best"].astype(np.int32), k)
        mpm = z[k + "_mpm"]
        bm = best[:, 0]
        w = np.where(bm == mpm[:, 0], 1, np.where((bm == mpm[:, 1]) | (bm == mpm[:, 2]), 2, 3))
        np.testing.assert_array_equal(best[:, 1], sads[np.arange(len(bm)), bm] + w * lam)
        for i in range(0, len(bm), 3):
            assert fnv1a(pred[i]) == z[k + "_fnv"][i, bm[i]], (k, i)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_intra_decide_luma_vs_oracle_random(gpu, oracle, n):
    """All four sizes on random inputs: noise / ramps / extreme samples, random most probable modes (repeats
    included), lambda 0 .. 51, CUs anywhere in the plane, one CU outside it (refused in its slot)."""
    rng = np.random.default_rng(900 + n)
    W, H, ncu = 160, 96, 48
    Y = rng.integers(0, 256, (H, W), dtype=np.uint8)
    Y[:, 80:] = (np.add.outer(np.arange(H), np.arange(80)) // 2 % 256).astype(np.uint8)
    xy = np.stack([rng.integers(0, W - n + 1, ncu), rng.integers(0, H - n + 1, ncu)], axis=1).astype(np.int32)
    xy[5] = (W - n + 1, 0)                                   # does not fit
    ref = rng.integers(0, 256, (ncu, 4 * n + 8), dtype=np.uint8)
    ref[::3] = np.clip(np.cumsum(rng.integers(-2, 3, ref[::3].shape), axis=1) + 120, 0, 255)
    ref[1::7] = rng.choice(np.array([0, 255], np.uint8), ref[1::7].shape)
    mpm = rng.integers(0, 35, (ncu, 3)).astype(np.uint8)
    mpm[::5, 1] = mpm[::5, 0]
    for lam in (0, 32, 51):
        best, sads, pred = gpu.intra_decide_luma(dev(Y), dev(xy), n, dev(ref), dev(mpm), lam, want_sads=True, want_pred=True)
        best, sads, pred = host(best), host(sads), host(pred)
        for i in range(ncu):
            if i == 5:
                assert best[i, 0] == -1 and best[i, 1] == -1
                continue
            x, y = xy[i]
            os_, obm, obc, opred = oracle.intra_decide_luma(np.ascontiguousarray(Y[y:y + n, x:x + n]),
                                                           np.ascontiguousarray(ref[i, :4 * n + 1]), np.ascontiguousarray(mpm[i]), lam)
            np.testing.assert_array_equal(sads[i].astype(np.uint32), os_, f"n {n} CU {i}")
            assert (best[i, 0], best[i, 1]) == (obm, obc), (n, i)
            np.testing.assert_array_equal(pred[i], opred)
    only_best, none_s, none_p = gpu.intra_decide_luma(dev(Y), dev(xy), n, dev(ref), dev(mpm), 51)
    assert none_s is None and none_p is None and np.array_equal(host(only_best), best)


# ---------------------------------------------------------- intra frame --
def _planes_of(yuv, w, h):
    y = yuv[:w * h].reshape(h, w)
    u = yuv[w * h:w * h + w * h // 4].reshape(h // 2, w // 2)
    v = yuv[w * h + w * h // 4:].reshape(h // 2, w // 2)
    return y, u, v


def test_intra_frame_cfg1_reconstruction_md5(gpu, oracle):
    """BASELINE.json configs[0] on the GPU: x265amd_intra_encode_frame (neighbour samples from the reconstruction,
    luma + chroma decisions, prediction, xSubDct / xQuant / xDeQuant / xIDctAdd, anti-diagonal by anti-diagonal)
    reconstructs every one of the 8 frames to the bytes whose MD5 the reference encoder printed
    (tests/golden/intra_cfg1.json), with the luma modes recorded from inside the reference on frames 0 and 5 and
    the levels / chroma modes of the oracle's restatement."""
    import hashlib, json, os
    from conftest import GOLDEN, load_golden
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    z, meta = load_golden("intra_decide.npz")
    w, h, cu, qp = doc["width"], doc["height"], doc["cu"], doc["qp"]
    yuv = synth.make_intra_probe(w, h, doc["frames"])
    for fr in range(doc["frames"]):
        y, u, v = _planes_of(yuv[fr], w, h)
        o = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp, wave_by_wave=(fr == 3))
        rec = np.concatenate([host(o["rec_y"]).ravel(), host(o["rec_u"]).ravel(), host(o["rec_v"]).ravel()])
        assert hashlib.md5(rec.tobytes()).hexdigest() == doc["recon_md5"][fr], fr
        for m in meta:
            if (m["width"], m["height"], m["cu"], m["qp"], m["frame"]) == (w, h, cu, qp, fr):
                np.testing.assert_array_equal(host(o["mode_y"]), z[m["key"] + "_best"])
        if fr in (0, 7):
            orec, my, mc, lv = oracle.intra_encode_frame(yuv[fr], w, h, cu, qp, want_levels=True)
            np.testing.assert_array_equal(host(o["mode_y"]), my)
            np.testing.assert_array_equal(host(o["mode_c"]), mc)
            lvl = np.concatenate([host(o["level_y"]).ravel(), host(o["level_u"]).ravel(), host(o["level_v"]).ravel()])
            np.testing.assert_array_equal(lvl, lv)


@pytest.mark.parametrize("w,h,cu,qp", [(416, 224, 32, 27), (128, 96, 32, 12), (96, 64, 16, 45), (32, 32, 32, 30), (16, 48, 16, 22),
                                        (32, 96, 32, 12), (64, 32, 32, 40), (32, 64, 16, 33)])
def test_intra_frame_vs_oracle(gpu, oracle, w, h, cu, qp):
    """Other sizes (CU 32, a single CU, a single column), quantisers and content against the oracle's restatement
    of the loop: reconstruction, modes, levels, xQuant sums."""
    from x265_amd import synth
    rng = np.random.default_rng(w + h + cu + qp)
    frames = [synth.make_intra_probe(w, h, 2)[1] if w % 2 == 0 else None,
              rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8),
              np.full(w * h * 3 // 2, 77, np.uint8)]
    for yuv in frames:
        y, u, v = _planes_of(yuv, w, h)
        o = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp)
        orec, my, mc, lv = oracle.intra_encode_frame(yuv, w, h, cu, qp, want_levels=True)
        rec = np.concatenate([host(o["rec_y"]).ravel(), host(o["rec_u"]).ravel(), host(o["rec_v"]).ravel()])
        np.testing.assert_array_equal(host(o["mode_y"]), my)
        np.testing.assert_array_equal(host(o["mode_c"]), mc)
        lvl = np.concatenate([host(o["level_y"]).ravel(), host(o["level_u"]).ravel(), host(o["level_v"]).ravel()])
        np.testing.assert_array_equal(lvl, lv)
        np.testing.assert_array_equal(rec, orec)
        ncx = w // cu
        sums = np.abs(lv[:w * h].reshape(h // cu, cu, ncx, cu).astype(np.int64)).sum(axis=(1, 3)).ravel()
        np.testing.assert_array_equal(host(o["abs_sum_y"]).astype(np.int64), sums)


def test_reference_bitstream_writer_over_gpu_intra_frames(gpu, reference):
    """The drop-in claim for intra pictures, end to end: the GPU computes the whole pixel path of all eight frames
    of BASELINE configs[0] (x265amd_intra_encode_frame: modes, levels, reconstruction), the REFERENCE's own bitstream
    writer (headers, most-probable-mode derivation, xWriteCU, CABAC; oracle/ref_shim.cpp ref_write_intra_frames =
    INTEGRATION.md's recipe) consumes them -- and the stream is the fixture's, byte for byte.  Needs oracle/_ref."""
    import hashlib, json, os
    from conftest import GOLDEN
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    w, h, cu, qp, nf = doc["width"], doc["height"], doc["cu"], doc["qp"], doc["frames"]
    yuv = synth.make_intra_probe(w, h, nf)
    rec, lvl, my, mc = [], [], [], []
    for fr in range(nf):
        y, u, v = _planes_of(yuv[fr], w, h)
        o = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp)
        rec.append(np.concatenate([host(o["rec_y"]).ravel(), host(o["rec_u"]).ravel(), host(o["rec_v"]).ravel()]))
        lvl.append(np.concatenate([host(o["level_y"]).ravel(), host(o["level_u"]).ravel(), host(o["level_v"]).ravel()]))
        my.append(host(o["mode_y"]))
        mc.append(host(o["mode_c"]))
    nbytes, bs = reference.write_intra_frames(yuv, w, h, cu, qp, np.stack(rec), np.stack(lvl), np.stack(my), np.stack(mc))
    assert [int(x) for x in nbytes] == doc["bytes_per_frame"]
    assert hashlib.md5(bs.tobytes()).hexdigest() == doc["bitstream_md5"]


@pytest.mark.parametrize("which", ["cfg1", "cu32"])
def test_gpu_intra_encoder_end_to_end_bitstream(gpu, which):
    """The intra encoder with nothing of the reference or the oracle in it: x265amd_intra_encode_frame (GPU pixel
    path) + x265amd_intra_write_picture (host syntax writer) on BASELINE configs[0] (CU 16) and on the CU-32
    encode of tests/golden/intra_stream.npz -- bytes per frame, the MD5 of the stream and the MD5 of every
    reconstruction are the ones the compiled reference produced."""
    import hashlib, json, os
    from conftest import GOLDEN, load_golden
    from x265_amd import synth
    if which == "cfg1":
        with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
            doc = json.load(f)
    else:
        doc = load_golden("intra_stream.npz")[1]["encode_cu32"]
    w, h, cu, qp, nf = doc["width"], doc["height"], doc["cu"], doc["qp"], doc["frames"]
    yuv = synth.make_intra_probe(w, h, nf)
    assert hashlib.md5(yuv.tobytes()).hexdigest() == doc["input_md5"]
    nbytes, parts = [], []
    for fr in range(nf):
        y, u, v = _planes_of(yuv[fr], w, h)
        o = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp)
        rec = np.concatenate([host(o["rec_y"]).ravel(), host(o["rec_u"]).ravel(), host(o["rec_v"]).ravel()])
        assert hashlib.md5(rec.tobytes()).hexdigest() == doc["recon_md5"][fr], fr
        bs = gpu.intra_write_picture(o["level_y"], o["level_u"], o["level_v"], o["mode_y"], o["mode_c"], cu, qp, fr)
        nbytes.append(len(bs))
        parts.append(bs)
    assert nbytes == doc["bytes_per_frame"]
    assert hashlib.md5(np.concatenate(parts).tobytes()).hexdigest() == doc["bitstream_md5"]


def test_intra_frames_batch_cfg1(gpu):
    """x265amd_intra_encode_frames: the eight frames of BASELINE configs[0] as ONE batch (anti-diagonal d of every
    frame in one launch) -- every reconstruction has the reference's MD5, and the batch's levels and modes through the
    syntax writer give the reference's bitstream."""
    import hashlib, json, os
    from conftest import GOLDEN
    from x265_amd import synth
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    w, h, cu, qp, nf = doc["width"], doc["height"], doc["cu"], doc["qp"], doc["frames"]
    yuv = synth.make_intra_probe(w, h, nf)
    o = gpu.intra_encode_frames(dev(yuv), w, h, cu, qp)
    rec, lvl, my, mc = host(o["rec"]), host(o["level"]), host(o["mode_y"]), host(o["mode_c"])
    parts = []
    for fr in range(nf):
        assert hashlib.md5(rec[fr].tobytes()).hexdigest() == doc["recon_md5"][fr], fr
        ly, lu, lv = lvl[fr, :w * h].reshape(h, w), lvl[fr, w * h:w * h * 5 // 4].reshape(h // 2, w // 2), lvl[fr, w * h * 5 // 4:].reshape(h // 2, w // 2)
        parts.append(gpu.intra_write_picture(ly, lu, lv, my[fr], mc[fr], cu, qp, fr))
    assert [len(p) for p in parts] == doc["bytes_per_frame"]
    assert hashlib.md5(np.concatenate(parts).tobytes()).hexdigest() == doc["bitstream_md5"]


@pytest.mark.parametrize("w,h,cu,qp,n", [(1920, 1088, 32, 30, 3), (416, 224, 32, 27, 5), (96, 64, 16, 45, 17), (32, 32, 32, 30, 2)])
def test_intra_frames_batch_equals_single_frames(gpu, oracle, w, h, cu, qp, n):
    """a batch == its frames one by one (x265amd_intra_encode_frame), on probe, noise and flat content; the small
    geometries also against the oracle.  Reusing the output buffers of a first call changes nothing."""
    from x265_amd import synth
    rng = np.random.default_rng(w * 3 + h + n)
    fsz = w * h * 3 // 2
    yuv = np.stack([synth.make_intra_probe(w, h, n)[n - 1]] + [rng.integers(0, 256, fsz, dtype=np.uint8) for _ in range(n - 2)] +
                   [np.full(fsz, 200, np.uint8)])
    o = gpu.intra_encode_frames(dev(yuv), w, h, cu, qp)
    o = gpu.intra_encode_frames(dev(yuv), w, h, cu, qp, out=o)
    for fr in range(n):
        y, u, v = _planes_of(yuv[fr], w, h)
        s = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp)
        rec = np.concatenate([host(s["rec_y"]).ravel(), host(s["rec_u"]).ravel(), host(s["rec_v"]).ravel()])
        lvl = np.concatenate([host(s["level_y"]).ravel(), host(s["level_u"]).ravel(), host(s["level_v"]).ravel()])
        np.testing.assert_array_equal(host(o["rec"][fr]), rec)
        np.testing.assert_array_equal(host(o["level"][fr]), lvl)
        for k in ("mode_y", "mode_c", "cost_y", "cost_c", "abs_sum_y", "abs_sum_u", "abs_sum_v"):
            np.testing.assert_array_equal(host(o[k][fr]), host(s[k]), k)
        if w * h <= 416 * 224:
            orec, my, mc, lv = oracle.intra_encode_frame(yuv[fr], w, h, cu, qp, want_levels=True)
            np.testing.assert_array_equal(host(o["rec"][fr]), orec)
            np.testing.assert_array_equal(host(o["level"][fr]), lv)


def test_intra_step_as_one_launch_equals_two_launches(gpu, oracle):
    """The transform stage at the end of the prediction workgroups (intra_wave_tq_kernel, the default) against the
    prediction kernel + transform launch pair (x265amd_debug_intra_fused(0)), and the 16-wave against the 8-wave shape
    of the kernel (x265amd_debug_intra_threads): every output equal, for a single frame (plane layout: U and V as two
    planes of one launch), a batch of frame buffers (plane pairs), CU 32 and CU 16."""
    from x265_amd import synth, api
    lib = api._lib()
    rng = np.random.default_rng(77)
    try:
        for (w, h, cu, qp, n) in ((416, 224, 32, 27, 3), (416, 240, 16, 32, 2), (1920, 1088, 32, 32, 2)):
            fsz = w * h * 3 // 2
            yuv = np.stack([synth.make_intra_probe(w, h, 1)[0]] + [rng.integers(0, 256, fsz, dtype=np.uint8) for _ in range(n - 1)])
            outs = []
            for fused, threads in ((1, 0), (0, 0), (1, 256), (1, 512), (1, 1024), (0, 256), (0, 512)):
                lib.x265amd_debug_intra_fused(fused)
                lib.x265amd_debug_intra_threads(threads)  # 0: by the step's size; else the 4- / 8- / 16-wave shape of the kernel
                o = gpu.intra_encode_frames(dev(yuv), w, h, cu, qp)
                y, u, v = _planes_of(yuv[n - 1], w, h)
                s = gpu.intra_encode_frame(dev(y), dev(u), dev(v), cu, qp)
                outs.append(({k: host(t) for k, t in o.items()}, {k: host(t) for k, t in s.items()}))
            for other in outs[1:]:
                for a, b in zip(outs[0], other):
                    assert a.keys() == b.keys()
                    for k in a:
                        np.testing.assert_array_equal(a[k], b[k], f"{w}x{h} cu {cu}: {k}")
            if w * h <= 416 * 240:
                orec, my, mc, lv = oracle.intra_encode_frame(yuv[1], w, h, cu, qp, want_levels=True)
                np.testing.assert_array_equal(outs[0][0]["rec"][1], orec)
                np.testing.assert_array_equal(outs[0][0]["level"][1], lv)
    finally:
        lib.x265amd_debug_intra_fused(1)
        lib.x265amd_debug_intra_threads(0)


@pytest.mark.parametrize("cu", [32, 16])
def test_intra_frames_plane_layouts(gpu, cu):
    """x265amd_intra_encode_frames through the C-ABI with the planes laid out three ways -- frame buffers Y,U,V (U and V
    of all frames as plane pairs of one transform launch), plane-major (all Y, all U, all V: one progression), and V
    before U (no common progression: the chroma planes take a launch each) -- gives the same planes, modes and sums."""
    import ctypes as C
    import torch
    from x265_amd import synth, api
    w, h, qp, n = 416, 224 if cu == 32 else 240, 30, 3
    ysz, csz = w * h, w * h // 4
    fsz, ncus = ysz + 2 * csz, (w // cu) * (h // cu)
    rng = np.random.default_rng(cu)
    yuv = np.stack([synth.make_intra_probe(w, h, 1)[0]] + [rng.integers(0, 256, fsz, dtype=np.uint8) for _ in range(n - 1)])
    want = {k: host(t) for k, t in gpu.intra_encode_frames(dev(yuv), w, h, cu, qp).items()}
    ys, us, vs = yuv[:, :ysz], yuv[:, ysz:ysz + csz], yuv[:, ysz + csz:]
    for order in ("yuv", "yvu"):
        first, second = (us, vs) if order == "yuv" else (vs, us)
        src = dev(np.concatenate([ys.ravel(), first.ravel(), second.ravel()]))
        bufs = {k: torch.zeros(n * fsz, dtype=dt, device=src.device) for k, dt in (("rec", torch.uint8), ("pred", torch.uint8), ("level", torch.int16))}
        small = {k: torch.zeros((n, ncus), dtype=torch.uint8, device=src.device) for k in ("mode_y", "mode_c")}
        sums = torch.zeros((3, n, ncus), dtype=torch.int32, device=src.device)  # Y, then the chroma planes in memory order
        costs = torch.zeros((2, n, ncus), dtype=torch.int32, device=src.device)
        f = api.IntraFrame()
        f.width, f.height, f.cu, f.qp, f.stride_y, f.stride_c = w, h, cu, qp, w, w // 2
        off_u = n * ysz + (0 if order == "yuv" else n * csz)
        off_v = n * ysz + (n * csz if order == "yuv" else 0)
        for name, t, esz in (("src", src, 1), ("rec", bufs["rec"], 1), ("pred", bufs["pred"], 1), ("level", bufs["level"], 2)):
            setattr(f, f"d_{name}_y", t.data_ptr())
            setattr(f, f"d_{name}_u", t.data_ptr() + esz * off_u)
            setattr(f, f"d_{name}_v", t.data_ptr() + esz * off_v)
        f.d_mode_y, f.d_mode_c = small["mode_y"].data_ptr(), small["mode_c"].data_ptr()
        f.d_cost_y, f.d_cost_c = costs[0].data_ptr(), costs[1].data_ptr()
        f.d_abs_sum_y = sums[0].data_ptr()
        f.d_abs_sum_u = sums[1 if order == "yuv" else 2].data_ptr()
        f.d_abs_sum_v = sums[2 if order == "yuv" else 1].data_ptr()
        api._check(api._lib().x265amd_intra_encode_frames(C.byref(f), n, ysz, csz, ncus, api._stream()), "x265amd_intra_encode_frames")
        for k in ("rec", "level"):
            got = host(bufs[k])
            gy = got[:n * ysz].reshape(n, ysz)
            gu = got[off_u:off_u + n * csz].reshape(n, csz)
            gv = got[off_v:off_v + n * csz].reshape(n, csz)
            np.testing.assert_array_equal(np.concatenate([gy, gu, gv], axis=1), want[k], f"{order}: {k}")
        for k in ("mode_y", "mode_c"):
            np.testing.assert_array_equal(host(small[k]), want[k], f"{order}: {k}")
        np.testing.assert_array_equal(host(costs[0]), want["cost_y"])
        np.testing.assert_array_equal(host(costs[1]), want["cost_c"])
        np.testing.assert_array_equal(host(sums[0]), want["abs_sum_y"])
        np.testing.assert_array_equal(host(sums[1 if order == "yuv" else 2]), want["abs_sum_u"], order)
        np.testing.assert_array_equal(host(sums[2 if order == "yuv" else 1]), want["abs_sum_v"], order)


def test_closed_loop_gop_with_intra_first_picture(gpu, oracle):
    """A whole GOP segment on the GPU: the I picture through the intra path (ClosedLoopGop.encode_i, the two segments
    as its batch), the P pictures searched against and predicted from ITS reconstruction.  CPU chain: the oracle's
    intra frame, then its search + TQ picture by picture."""
    import hashlib
    from x265_amd import synth, ClosedLoopGop
    W, H, R, QP, T, CU = 256, 128, 16, 30, 3, 32
    segs = []
    for seed in (15, 16):
        y, u, v = synth.make_sequence(W, H, T, search_range=R, seed=seed, chroma=True)
        segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
    yuv = np.stack(segs)
    S, ysz, csz = 2, W * H, (W // 2) * (H // 2)
    enc = ClosedLoopGop(W, H, S, search_range=R, qp=QP)
    rec0 = host(enc.encode_i(dev(yuv[:, 0]), cu=CU))
    ref_cpu = []
    for s in range(S):
        orec, my, mc, lv = oracle.intra_encode_frame(yuv[s, 0], W, H, CU, QP, want_levels=True)
        np.testing.assert_array_equal(rec0[s], orec)
        np.testing.assert_array_equal(host(enc.level)[s], lv)
        np.testing.assert_array_equal(host(enc.mode_y)[s], my)
        np.testing.assert_array_equal(host(enc.mode_c)[s], mc)
        ref_cpu.append(orec.copy())
    qpc = oracle.chroma_qp(QP)
    for t in range(1, T):
        rec_gpu = host(enc.encode_p(dev(yuv[:, t])))
        gmv = gpu.mv_to_numpy(enc.mv)[:, 0]
        for s in range(S):
            cur, ref = yuv[s, t], ref_cpu[s]
            cy, ry = cur[:ysz].reshape(H, W), ref[:ysz].reshape(H, W)
            omv = oracle.me_fullsearch(cy, ry, 64, R, nthreads=8)
            np.testing.assert_array_equal(gmv[s], omv)
            ly, _ = oracle.tq_forward(cy, ry, 32, QP, 1, mv=omv, mv_ctu=64)
            rec = [oracle.tq_inverse(ly, ry, 32, QP, 1, mv=omv, mv_ctu=64).reshape(-1)]
            cmv = omv.copy()
            cmv["mvx"] >>= 1
            cmv["mvy"] >>= 1
            for k in range(2):
                a, b = ysz + k * csz, ysz + (k + 1) * csz
                cc, rc = cur[a:b].reshape(H // 2, W // 2), ref[a:b].reshape(H // 2, W // 2)
                lc, _ = oracle.tq_forward(cc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32)
                rec.append(oracle.tq_inverse(lc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32).reshape(-1))
            ref_cpu[s] = np.concatenate(rec)
            assert hashlib.md5(rec_gpu[s].tobytes()).hexdigest() == hashlib.md5(ref_cpu[s].tobytes()).hexdigest(), (s, t)
    assert enc.poc == T - 1


def test_me_sea_32x32_blocks_1080p(gpu, oracle):
    """The elimination search on a 32x32 PU grid (the same kernels at half the size, 8x8 sub-block bounds) at
    BASELINE configs[1]'s geometry, two frames batched: every vector and cost equals the every-candidate kernel's,
    a sample of blocks equals the oracle's, and the workspace can be reused."""
    from x265_amd import synth, api
    W, H, R = 1920, 1088, 64
    seq = synth.make_sequence(W, H, 3, search_range=R, seed=12345)
    cur, ref = dev(np.ascontiguousarray(seq[1:])), dev(np.ascontiguousarray(seq[:-1]))
    ws = torch.empty(api.me_workspace_bytes(W, H, R, 2, 1, 32), dtype=torch.uint8, device="cuda")
    a = gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 32, R, method="brute"))
    for _ in range(2):
        b = gpu.mv_to_numpy(gpu.me_fullsearch(cur, ref, 32, R, method="sea", workspace=ws))
        np.testing.assert_array_equal(a, b)
    for f in range(2):
        want = oracle.me_fullsearch(seq[f + 1], seq[f], 32, R, first=5 + f, step=211, nthreads=8)   # every 211th block
        idx = np.arange(5 + f, a.shape[2], 211)
        np.testing.assert_array_equal(a[f, 0][idx], want[idx], f"frame {f}")


def test_intra_api_rejects_wrong_output_buffers(gpu):
    """a caller's `out` of another shape, type or layout is refused before anything is launched (the library writes
    through those pointers)"""
    from x265_amd import api, synth
    w, h, cu = 64, 32, 16
    yuv = dev(synth.make_intra_probe(w, h, 2))
    good = gpu.intra_encode_frames(yuv, w, h, cu, 30)
    for k, bad in (("level", good["level"].to(torch.int32)), ("rec", good["rec"][:, :-1]), ("mode_y", good["mode_y"][:1]),
                   ("pred", good["pred"].t().contiguous().t())):
        with pytest.raises(ValueError):
            gpu.intra_encode_frames(yuv, w, h, cu, 30, out=dict(good, **{k: bad}))
    with pytest.raises(ValueError):
        gpu.intra_encode_frames(yuv, w, h, cu, 30, out={k: v for k, v in good.items() if k != "cost_c"})
    y, u, v = yuv[0, :w * h].view(h, w), yuv[0, w * h:w * h * 5 // 4].view(h // 2, w // 2), yuv[0, w * h * 5 // 4:].view(h // 2, w // 2)
    one = api.intra_frame_buffers(h, w, cu, y.device)
    with pytest.raises(ValueError):
        gpu.intra_encode_frame(y.contiguous(), u.contiguous(), v.contiguous(), cu, 30, out=dict(one, level_u=one["level_y"]))


def test_me_random_geometries(gpu, oracle):
    """tools/me_fuzz.py, 16 cases: random picture sizes, block sides 64 / 32, ranges 1 .. 128, frames x references, row
    strides wider than the picture, noise / flat / shifted content -- the elimination search and the one-candidate-
    per-lane kernel equal the every-candidate kernel, and a sample of blocks equals the CPU oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("me_fuzz", os.path.join(ROOT, "tools", "me_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["me_fuzz.py", "16", "3"]
        assert mod.main() == 0
    finally:
        sys.argv = argv


def test_tq_random_geometries(gpu, oracle):
    """tools/tq_fuzz.py, 24 cases: random plane sizes (multiples of the TU, not of the 32x32 work tile), TU 4 .. 32, QP
    0 .. 51, all three roundings, odd row strides, extreme residuals, per-TU modes at 4x4 -- forward, inverse and the
    fused round trip agree with each other and with the CPU oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tq_fuzz", os.path.join(ROOT, "tools", "tq_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["tq_fuzz.py", "24", "9"]
        assert mod.main() == 0
    finally:
        sys.argv = argv


def test_intra_random_geometries(gpu, oracle):
    """tools/intra_fuzz.py, 12 cases: random picture sizes (one CU wide or high included), CU 16 / 32, QP 0 .. 51,
    batches of 1 .. 5 frames of noise, flat, smooth and blocky content -- reconstruction, levels and modes of
    x265amd_intra_encode_frames equal the oracle's restatement of the loop."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("intra_fuzz", os.path.join(ROOT, "tools", "intra_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["intra_fuzz.py", "12", "3"]
        assert mod.main() == 0
    finally:
        sys.argv = argv


def test_pu_random_geometries(gpu, oracle):
    """tools/pu_fuzz.py, 8 cases: the square-PU tree and the all-shapes search on random picture sizes, ranges 1 .. 128,
    frames x references and content -- squares 64 / 32 / 16 equal the every-candidate search at that block size, the
    rectangles and 8x8 equal the CPU oracle on a sample of blocks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pu_fuzz", os.path.join(ROOT, "tools", "pu_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["pu_fuzz.py", "8", "4"]
        assert mod.main() == 0
    finally:
        sys.argv = argv


def test_gop_random_geometries(gpu, oracle):
    """tools/gop_fuzz.py, 5 cases: whole GOP segments (the I picture through the intra path, two P pictures in closed
    loop, luma and chroma) on random picture sizes, ranges, quantisers, segment counts and intra CU sizes against the
    same chain on the CPU oracle: vectors, levels and every reconstructed picture."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gop_fuzz", os.path.join(ROOT, "tools", "gop_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["gop_fuzz.py", "5", "8"]
        assert mod.main() == 0
    finally:
        sys.argv = argv
