"""SATD (sum of absolute Hadamard-transformed differences).  The reference has none:
PARITY UNPINNED -- the definition is the HEVC test model's published one (xCalcHADs8x8 / xCalcHADs4x4); the
CPU tests pin the oracle's matrix restatement to closed-form cases and to an independent numpy Hadamard, the
GPU tests compare the HIP kernel with the oracle, bit-exact."""
import numpy as np
import pytest

SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 16), (16, 8), (64, 32), (32, 64), (4, 64)]


def hadamard(m):
    h = np.array([[1]])
    while h.shape[0] < m:
        h = np.block([[h, h], [h, -h]])
    return h


def satd_numpy(w, n, a, b):
    m = 4 if (w == 4 or n == 4) else 8
    h = hadamard(m)
    d = a.astype(np.int64) - b.astype(np.int64)
    tot = 0
    for y in range(0, n, m):
        for x in range(0, w, m):
            s = int(np.abs(h @ d[y:y + m, x:x + m] @ h.T).sum())
            tot += (s + 2) >> 2 if m == 8 else (s + 1) >> 1
    return tot


def test_oracle_satd_closed_forms(oracle):
    """A constant difference d leaves only the DC coefficient: 64|d| per 8x8 -> 16|d|, 16|d| per 4x4 -> 8|d|.
    A single differing pixel spreads |d| over every coefficient: 64|d| -> 16|d| again."""
    for w, n in SIZES:
        m = 4 if (w == 4 or n == 4) else 8
        nsub = (w // m) * (n // m)
        a = np.full(w * n, 200, np.uint8)
        b = np.full(w * n, 187, np.uint8)
        per = (m * m * 13 + (2 if m == 8 else 1)) >> (2 if m == 8 else 1)
        assert oracle.satd(w, n, a, w, b, w) == nsub * per
        assert oracle.satd(w, n, a, w, a, w) == 0
    a = np.zeros(64, np.uint8)
    b = a.copy()
    b[27] = 255
    assert oracle.satd(8, 8, a, 8, b, 8) == (64 * 255 + 2) >> 2
    assert oracle.satd(8, 8, b, 8, a, 8) == (64 * 255 + 2) >> 2      # symmetric


def test_oracle_satd_vs_numpy_hadamard(oracle):
    rng = np.random.default_rng(3)
    for w, n in SIZES:
        for kind in ("random", "extreme"):
            ss, sr = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 17))
            a = rng.integers(0, 256, n * ss, dtype=np.uint8)
            b = rng.integers(0, 256, n * sr, dtype=np.uint8)
            if kind == "extreme":          # checkerboard of +-255: the largest coefficient an 8x8 can hold
                a.reshape(n, ss)[:, :w] = ((np.add.outer(np.arange(n), np.arange(w)) & 1) * 255).astype(np.uint8)
                b.reshape(n, sr)[:, :w] = 255 - a.reshape(n, ss)[:, :w]
            want = satd_numpy(w, n, a.reshape(n, ss)[:, :w], b.reshape(n, sr)[:, :w])
            assert oracle.satd(w, n, a, ss, b, sr) == want, (w, n, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("w,n", SIZES)
def test_satd_batch_vs_oracle(gpu, oracle, w, n):
    """Many blocks at arbitrary (unaligned) offsets of two planes with different strides."""
    import torch
    rng = np.random.default_rng(w * 100 + n)
    ss, sr, rows = 200, 312, 160
    src = rng.integers(0, 256, rows * ss, dtype=np.uint8)
    ref = rng.integers(0, 256, rows * sr, dtype=np.uint8)
    src[:ss * 70] = 255        # a band of extremes: |d| = 255 everywhere some blocks land
    ref[:sr * 70] = 0
    nb = 300
    so = (rng.integers(0, rows - 64, nb) * ss + rng.integers(0, ss - 64, nb)).astype(np.int32)
    ro = (rng.integers(0, rows - 64, nb) * sr + rng.integers(0, sr - 64, nb)).astype(np.int32)
    dev = lambda a: torch.from_numpy(a).cuda()
    got = gpu.satd_blocks(w, n, dev(src), ss, dev(ref), sr, dev(so), dev(ro)).cpu().numpy()
    want = [oracle.satd(w, n, src[so[b]:], ss, ref[ro[b]:], sr) for b in range(nb)]
    np.testing.assert_array_equal(got.astype(np.uint32), np.array(want, np.uint32))


@pytest.mark.gpu
def test_satd_checkerboard_and_errors(gpu):
    """The worst case for the packed Int16 butterflies (every |d| = 255, alternating sign) and argument checks."""
    import torch
    from x265_amd.api import X265AmdError
    a = ((np.add.outer(np.arange(64), np.arange(64)) & 1) * 255).astype(np.uint8)
    b = (255 - a).astype(np.uint8)
    off = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = gpu.satd_blocks(64, 64, torch.from_numpy(a).cuda(), 64, torch.from_numpy(b).cuda(), 64, off, off)
    assert int(got[0]) == 64 * ((64 * 255 + 2) >> 2)       # one coefficient of 64 * 255 per 8x8
    with pytest.raises(X265AmdError):
        gpu.satd_blocks(12, 8, torch.from_numpy(a).cuda(), 64, torch.from_numpy(b).cuda(), 64, off, off)
