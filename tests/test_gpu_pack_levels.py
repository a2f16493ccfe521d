"""GPU: x265amd_pack_levels (the compaction of level planes for the host) against the dense planes it packs.

The packed form carries exactly what the dense plane does: unpacking it gives the plane back, bit for bit, the table's
counts are the numbers of non-zero levels per TU, and the TU's words are in raster order (what the host's coefficient
walk, src/encode.cpp:270-461, wants).  Run on the MI355X box:  python -u -m pytest tests -m gpu -x -v
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sparse_levels(rng, f, h, w, density, full_range=False):
    lv = np.zeros((f, h, w), np.int16)
    mask = rng.random((f, h, w)) < density
    if full_range:
        vals = rng.integers(-32768, 32768, (f, h, w)).astype(np.int16)
    else:
        vals = rng.integers(-40, 41, (f, h, w)).astype(np.int16)
    lv[mask] = vals[mask]
    return lv


def _check(gpu, lv, tu, table, words, cursor):
    f, h, w = lv.shape
    tb = table.cpu().numpy()
    n = int(cursor.cpu().numpy()[0])
    assert n == int(np.count_nonzero(lv))
    wd = words.cpu().numpy()[:n]
    got = gpu.unpack_levels(tb, wd, w, h, tu)
    np.testing.assert_array_equal(got, lv)
    # counts per TU, room of the TUs disjoint and covering [0, n), raster order inside a TU
    blocks = lv.reshape(f, h // tu, tu, w // tu, tu).transpose(0, 1, 3, 2, 4).reshape(f, -1, tu * tu)
    np.testing.assert_array_equal(tb[..., 1], np.count_nonzero(blocks, axis=2))
    used = tb.reshape(-1, 2)
    used = used[used[:, 1] > 0]
    if n == 0:
        assert used.size == 0
        return
    order = np.argsort(used[:, 0])
    ends = used[order, 0] + used[order, 1]
    assert used[order[0], 0] == 0 and ends[-1] == n
    np.testing.assert_array_equal(used[order[1:], 0], ends[:-1])
    for off, cnt in used[:50]:
        pos = wd[off:off + cnt].view(np.uint32) >> 16
        assert np.all(np.diff(pos.astype(np.int64)) > 0)


@pytest.mark.parametrize("tu", [4, 8, 16, 32])
@pytest.mark.parametrize("density", [0.0, 0.02, 0.5, 1.0])
def test_pack_levels_gives_the_plane_back(gpu, tu, density):
    rng = np.random.default_rng(tu * 100 + int(density * 50))
    lv = _sparse_levels(rng, 3, 96, 160, density, full_range=density == 1.0)
    if density > 0:
        lv[0, 0, 0] = 7  # something at offset 0
    d = torch.from_numpy(lv).cuda()
    table, words, cursor = gpu.pack_levels(d, tu)
    if density == 0.0:
        assert int(cursor.cpu()[0]) == 0 and not table.cpu().numpy().any()
        return
    _check(gpu, lv, tu, table, words, cursor)


@pytest.mark.parametrize("tu", [4, 8, 16, 32])
def test_pack_levels_after_the_transform_stage(gpu, tu):
    """Levels and per-TU sums as x265amd_tq_forward leaves them; a padded plane (row stride > width), one frame."""
    rng = np.random.default_rng(tu)
    h, w = 128, 192
    src = torch.from_numpy(rng.integers(0, 256, (2, h, w), dtype=np.uint8)).cuda()
    noise = rng.integers(-25, 26, (2, h, w))
    noise[:, :, :w // 2] = 0  # the left half predicts perfectly: TUs without a level
    pred = torch.from_numpy(np.clip(src.cpu().numpy().astype(np.int64) + noise, 0, 255).astype(np.uint8)).cuda()
    level, sums = gpu.tq_forward(src, pred, tu, 30, gpu.SLICE_P)
    lv = level.cpu().numpy()
    assert np.count_nonzero(lv) > 0 and np.count_nonzero(sums.cpu().numpy() == 0) > 0  # coded and uncoded TUs both
    with_sums = gpu.pack_levels(level, tu, abs_sum=sums)
    _check(gpu, lv, tu, *with_sums)
    without = gpu.pack_levels(level, tu)
    np.testing.assert_array_equal(with_sums[0].cpu().numpy()[..., 1], without[0].cpu().numpy()[..., 1])
    # ragged stride: the plane as the left part of a wider one
    wide = torch.zeros((2, h, w + 40), dtype=torch.int16, device="cuda")
    wide[:, :, :w] = level
    wide[:, :, w:] = 99  # never read
    t2 = gpu.pack_levels(wide[:, :, :w], tu, abs_sum=sums)
    _check(gpu, lv, tu, *t2)
    # rows that start on odd 2-byte boundaries: the path without the 16-byte loads
    odd = torch.zeros((2, h, w + 3), dtype=torch.int16, device="cuda")
    odd[:, :, 1:w + 1] = level
    _check(gpu, lv, tu, *gpu.pack_levels(odd[:, :, 1:w + 1], tu, abs_sum=sums))


def test_pack_levels_planes_share_one_cursor(gpu):
    """Luma then chroma into the same words: begin once, three calls (what the host's fetch does)."""
    rng = np.random.default_rng(5)
    y = _sparse_levels(rng, 2, 128, 128, 0.1)
    u = _sparse_levels(rng, 2, 64, 64, 0.05)
    dy, du = torch.from_numpy(y).cuda(), torch.from_numpy(u).cuda()
    words = torch.empty(y.size + u.size, dtype=torch.int32, device="cuda")
    cursor = torch.zeros(1, dtype=torch.int32, device="cuda")
    ty = torch.empty((2, 16, 2), dtype=torch.int32, device="cuda")
    tu_ = torch.empty((2, 16, 2), dtype=torch.int32, device="cuda")
    gpu.pack_levels(dy, 32, out=(ty, words, cursor))
    gpu.pack_levels(du, 16, out=(tu_, words, cursor), begin=False)
    n = int(cursor.cpu()[0])
    assert n == np.count_nonzero(y) + np.count_nonzero(u)
    wd = words.cpu().numpy()[:n]
    np.testing.assert_array_equal(gpu.unpack_levels(ty.cpu().numpy(), wd, 128, 128, 32), y)
    np.testing.assert_array_equal(gpu.unpack_levels(tu_.cpu().numpy(), wd, 64, 64, 16), u)


def test_pack_levels_capacity_and_arguments(gpu):
    """Past the capacity nothing is written out of bounds and the cursor still says what was needed."""
    rng = np.random.default_rng(9)
    lv = _sparse_levels(rng, 1, 64, 64, 0.5)
    d = torch.from_numpy(lv).cuda()
    need = int(np.count_nonzero(lv))
    words = torch.full((need,), -1, dtype=torch.int32, device="cuda")
    cursor = torch.zeros(1, dtype=torch.int32, device="cuda")
    table = torch.empty((1, 64, 2), dtype=torch.int32, device="cuda")
    cap = need // 2
    gpu.pack_levels(d, 8, capacity=cap, out=(table, words, cursor))
    assert int(cursor.cpu()[0]) == need
    assert (words.cpu().numpy()[cap:] == -1).all()
    tb = table.cpu().numpy()[0]
    wd = words.cpu().numpy()
    for t, (off, cnt) in enumerate(tb):  # TUs that fitted are whole and right
        if cnt and off + cnt <= cap:
            ty, tx = divmod(t, 8)
            blk = lv[0, ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(-1)
            nz = np.flatnonzero(blk)
            np.testing.assert_array_equal(wd[off:off + cnt].view(np.uint32) >> 16, nz)
            np.testing.assert_array_equal((wd[off:off + cnt] & 0xffff).astype(np.uint16).view(np.int16), blk[nz])
    # the table / the words in ordinary host memory: an error code, not a GPU fault
    from x265_amd import api
    host_words = np.zeros(need, np.int32)
    st = api._lib().x265amd_pack_levels(d.data_ptr(), 64, 64, 64, 8, 1, 0, None, 0, table.data_ptr(), host_words.ctypes.data, need,
                                        cursor.data_ptr(), None)
    assert st == -1, st  # X265AMD_ERR_ARG
    with pytest.raises(gpu.X265AmdError):
        gpu.pack_levels(d, 12)
    with pytest.raises(gpu.X265AmdError):
        gpu.pack_levels(torch.zeros((1, 60, 64), dtype=torch.int16, device="cuda"), 8)


def test_copy_to_host_by_kernel(gpu):
    """x265amd_copy_to_host: device memory into page-locked host memory by a kernel of the stream; anything that is not
    page-locked host memory is refused."""
    import ctypes as C
    from x265_amd import api
    lib = api._lib()
    for n in (4, 20, 4096, 1 << 20):  # bytes: the 4-byte path (20) and the 16-byte one
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
        hp = C.c_void_p()
        assert lib.x265amd_malloc_host(C.byref(hp), n) == 0
        try:
            assert lib.x265amd_copy_to_host(hp, src.data_ptr(), n, None) == 0
            torch.cuda.synchronize()
            got = np.frombuffer((C.c_uint8 * n).from_address(hp.value), np.uint8)
            np.testing.assert_array_equal(got, src.cpu().numpy())
        finally:
            lib.x265amd_free_host(hp)
    pageable = np.zeros(64, np.uint8)
    src = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert lib.x265amd_copy_to_host(pageable.ctypes.data, src.data_ptr(), 64, None) == -1
    assert lib.x265amd_copy_to_host(pageable.ctypes.data, src.data_ptr(), 6, None) == -1  # not a multiple of 4
