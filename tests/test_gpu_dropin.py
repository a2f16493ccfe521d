"""GPU parity of the per-block drop-ins (include/x265_amd_dropin.h): the reference's own signatures
(inc/x265.h:290-298) on host buffers, computed by the HIP kernels.  The cases are the ones the compiled
reference produced for tests/golden/primitives.npz, then seeded random blocks against the CPU oracle's
per-block functions -- call for call what a CHECK_TV run of the reference would compare.  Bit-exact."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MODE_INVALID = 255


@pytest.fixture(scope="module")
def dropin(gpu):
    from x265_amd import dropin
    return dropin


def test_dropin_sad_golden(dropin):
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "sad"]
    assert len(cases) == 30
    for m in cases:
        k = m["id"]
        idx = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}[m["w"]]            # xSadN[xLog2(w-1) - 2]
        got = dropin.xSadN[idx](m["n"], z[f"sad{k}_src"], m["ss"], z[f"sad{k}_ref"], m["sr"])
        assert got == int(z[f"sad{k}_val"]), m


def test_dropin_1d_passes_golden(dropin):
    """xDctN[] / xInvDctN[] one pass at a time, with the production shifts (the fixture's `t1d` cases)."""
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "t1d"]
    assert len(cases) == 20
    for m in cases:
        k = m["id"]
        x = z[f"t1d{k}_in"]
        dst = np.zeros_like(x)
        table = dropin.xDctN if m["dir"] == "fwd" else dropin.xInvDctN
        table[m["idx"]](dst, x, 32, m["n"], m["shift"])
        np.testing.assert_array_equal(dst, z[f"t1d{k}_out"], str(m))


def test_dropin_tq_chain_golden(dropin):
    """xSubDct -> xQuant -> xDeQuant -> xIDctAdd, each stage against the reference's output for that stage."""
    z, meta = load_golden("primitives.npz")
    cases = [m for m in meta if m["kind"] == "tq"]
    assert len(cases) > 150
    for m in cases:
        k, n = m["id"], m["n"]
        src, pred = z[f"tq{k}_src"], z[f"tq{k}_pred"]
        t0 = np.zeros(n * 32, np.int16)
        t1 = np.zeros(n * 32, np.int16)
        dropin.xSubDct(t0, src, pred, 32, t0, t1, n, n, m["mode"])     # pDst aliases piTmp0: src/encode.cpp:868-872
        np.testing.assert_array_equal(t0, z[f"tq{k}_coef"], f"coef {m}")
        level = np.zeros_like(t0)
        s = dropin.xQuant(level, t0, 32, m["qp"], n, n, m["slice"])
        np.testing.assert_array_equal(level, z[f"tq{k}_level"], f"level {m}")
        assert s == int(z[f"tq{k}_sum"]), m
        deq = np.zeros_like(t0)
        dropin.xDeQuant(deq, level, 32, m["qp"], n, n, m["slice"])
        np.testing.assert_array_equal(deq, z[f"tq{k}_deq"], f"deq {m}")
        rec = np.zeros(n * 32, np.uint8)
        dropin.xIDctAdd(rec, deq, pred, 32, t0, t1, n, n, m["mode"])
        np.testing.assert_array_equal(rec, z[f"tq{k}_rec"], f"rec {m}")


@pytest.mark.parametrize("idx,n", [(0, 4), (1, 4), (2, 8), (3, 16), (4, 32)])
def test_dropin_1d_vs_oracle_any_lines(dropin, oracle, idx, n):
    """Line counts other than N, strides other than 32, the full Int16 range, every legal shift."""
    rng = np.random.default_rng(100 + idx)
    for lines, stride, shift in ((1, 40, 1), (5, 33, 7), (n, n, 12), (37, 48, 9), (64, 64, 15)):
        stride = max(stride, n, lines)
        for direction in ("fwd", "inv"):
            if direction == "inv" and n == 32:
                stride = max(stride, 32)       # xInvDCT32 runs 32 lines whatever nLines says (src/pixel.cpp:513)
            rows = max(n, lines)
            x = rng.integers(-32768, 32768, rows * stride).astype(np.int16)
            x[:stride] = 32767
            x[stride:2 * stride] = -32768
            want = (oracle.fwd_1d if direction == "fwd" else oracle.inv_1d)(idx, x, stride, lines, shift)
            dst = np.zeros_like(x)
            (dropin.xDctN if direction == "fwd" else dropin.xInvDctN)[idx](dst, x, stride, lines, shift)
            np.testing.assert_array_equal(dst, want, f"{direction} idx {idx} lines {lines} stride {stride} shift {shift}")


def test_dropin_dst4_ignores_line_count(dropin, oracle):
    """xDST4 / xInvDST4 loop over four lines whatever nLines says (src/pixel.cpp:138, 351)."""
    rng = np.random.default_rng(5)
    x = rng.integers(-255, 256, 4 * 8).astype(np.int16)
    for table, f in ((dropin.xDctN, oracle.fwd_1d), (dropin.xInvDctN, oracle.inv_1d)):
        dst = np.zeros_like(x)
        table[0](dst, x, 8, 2, 7)
        np.testing.assert_array_equal(dst, f(0, x, 8, 2, 7))
        assert dst.reshape(4, 8)[:, :4].any() and np.array_equal(dst, f(0, x, 8, 4, 7))


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_dropin_blocks_vs_oracle_random_strides(dropin, oracle, n):
    rng = np.random.default_rng(n)
    for trial in range(12):
        stride = n + int(rng.integers(0, 40))
        mode = int(rng.choice([MODE_INVALID, 0, 1, 5, 12, 26, 34]))
        qp = int(rng.integers(0, 52))
        st = int(rng.integers(0, 3))
        src = rng.integers(0, 256, n * stride, dtype=np.uint8)
        pred = np.clip(src.astype(np.int32) + rng.integers(-40, 41, src.size), 0, 255).astype(np.uint8) if trial & 1 \
            else rng.integers(0, 256, n * stride, dtype=np.uint8)
        coef = np.full(n * stride, 77, np.int16)
        dropin.xSubDct(coef, src, pred, stride, None, None, n, n, mode)
        want = oracle.sub_dct(src, pred, stride, n, n, mode)
        blk = lambda a: a.reshape(n, stride)[:, :n]
        np.testing.assert_array_equal(blk(coef), blk(want))
        assert (coef.reshape(n, stride)[:, n:] == 77).all(), "wrote outside the block"
        level = np.zeros_like(coef)
        s = dropin.xQuant(level, coef, stride, qp, n, n, st)
        wl, ws = oracle.quant(want, stride, qp, n, n, st)
        np.testing.assert_array_equal(blk(level), blk(wl))
        assert s == ws
        # the inverse chain on arbitrary Int16 coefficients (clipping / wrapping paths of App. C)
        wild = rng.integers(-32768, 32768, n * stride).astype(np.int16)
        deq = np.zeros_like(wild)
        dropin.xDeQuant(deq, wild, stride, qp, n, n, st)
        np.testing.assert_array_equal(blk(deq), blk(oracle.dequant(wild, stride, qp, n, n, st)))
        rec = np.zeros(n * stride, np.uint8)
        dropin.xIDctAdd(rec, wild, pred, stride, None, None, n, n, mode)
        np.testing.assert_array_equal(blk(rec), blk(oracle.idct_add(wild, pred, stride, n, n, mode)))


def test_dropin_idct_add_plane_vs_oracle(gpu, oracle):
    """x265amd_idct_add (xIDctAdd alone) on a whole ragged plane of dequantised coefficients."""
    import torch
    rng = np.random.default_rng(9)
    for n in (4, 8, 16, 32):
        h, w = 3 * n, 5 * n
        coef = rng.integers(-3000, 3001, (h, w)).astype(np.int16)
        pred = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rec = gpu.idct_add(torch.from_numpy(coef).cuda(), torch.from_numpy(pred).cuda(), n)[0].cpu().numpy()
        for ty in range(h // n):
            for tx in range(w // n):
                c = np.zeros(n * 32, np.int16)
                p = np.zeros(n * 32, np.uint8)
                c.reshape(n, 32)[:, :n] = coef[ty * n:(ty + 1) * n, tx * n:(tx + 1) * n]
                p.reshape(n, 32)[:, :n] = pred[ty * n:(ty + 1) * n, tx * n:(tx + 1) * n]
                want = oracle.idct_add(c, p, 32, n, n).reshape(n, 32)[:, :n]
                np.testing.assert_array_equal(rec[ty * n:(ty + 1) * n, tx * n:(tx + 1) * n], want)


def test_dropin_rejects_what_the_reference_asserts_on(dropin, gpu):
    """TU > 32 is an assert in the reference (src/pixel.cpp:684); here a status, and nothing is written."""
    from x265_amd.api import X265AmdError
    a = np.zeros(64 * 64, np.uint8)
    out = np.full(64 * 64, 5, np.int16)
    with pytest.raises(X265AmdError) as e:
        dropin.xSubDct(out, a, a, 64, None, None, 64, 64)
    assert e.value.status == -1 and (out == 5).all()
    with pytest.raises(X265AmdError):
        dropin.xQuant(out, out.copy(), 8, 52, 8, 8, 0)
    with pytest.raises(X265AmdError):
        dropin.xSadN[0](0, a, 4, a, 4)
    assert dropin.xSadN[4](64, a, 64, a, 64) == 0        # and the layer still works afterwards


def test_reference_encoder_over_gpu_pixel_tables(gpu, reference):
    """The drop-in claim end to end: the REAL reference encoder (oracle/_ref, compiled from the reference's
    sources) with its three mutable plug-in tables xSadN[] / xDctN[] / xInvDctN[] (src/pixel.cpp:95,571,579)
    pointed at the GPU drop-ins encodes BASELINE configs[0] (all 8 frames) to the same bitstream, byte for
    byte (MD5 of the stream), and the same reconstruction MD5s as the committed fixture of the all-CPU run.  Every SAD of the intra mode
    decision and every 1-D transform pass of xSubDct / xIDctAdd goes through the GPU, one block per call."""
    import ctypes as C
    import json
    import os

    from conftest import GOLDEN
    from x265_amd import api, synth
    lib = api.load_library()
    addr = lambda name: C.cast(getattr(lib, name), C.c_void_p).value
    sad = [addr(f"x265amd_xSad{w}xN") for w in (4, 8, 16, 32, 64)]
    dct = [addr("x265amd_" + n) for n in ("xDST4", "xDCT4", "xDCT8", "xDCT16", "xDCT32")]
    inv = [addr("x265amd_" + n) for n in ("xInvDST4", "xInvDCT4", "xInvDCT8", "xInvDCT16", "xInvDCT32")]
    with open(os.path.join(GOLDEN, "intra_cfg1.json")) as f:
        doc = json.load(f)
    import hashlib
    nf = doc["frames"]
    yuv = synth.make_intra_probe(doc["width"], doc["height"], nf)
    saved = reference.get_tables()
    calls0 = lib.x265amd_dropin_calls()
    try:
        reference.set_tables(sad, dct, inv)
        assert reference.get_tables() == [sad, dct, inv]
        rec, nbytes, md5, bs = reference.encode_intra(yuv, doc["width"], doc["height"], doc["cu"], doc["qp"])
    finally:
        reference.set_tables(*saved)
    served = lib.x265amd_dropin_calls() - calls0
    ncu = (doc["width"] // doc["cu"]) * (doc["height"] // doc["cu"])
    assert served > nf * ncu * 20, f"only {served} calls reached the GPU"
    assert lib.x265amd_dropin_status() == 0
    assert [int(x) for x in nbytes] == doc["bytes_per_frame"][:nf]
    assert [bytes(m).hex() for m in md5] == doc["recon_md5"][:nf]
    assert hashlib.md5(bs.tobytes()).hexdigest() == doc["bitstream_md5"]
    # and the tables are the reference's own again
    cpu = reference.encode_intra(yuv[:1], doc["width"], doc["height"], doc["cu"], doc["qp"])
    assert lib.x265amd_dropin_calls() - calls0 == served and bytes(cpu[2][0]).hex() == doc["recon_md5"][0]
