"""GPU parity at the FULL sizes of BASELINE.json configs[3] and configs[4] in the batch shapes bench.py
runs (frames x references in one launch), plus the 1080p chroma plane of the closed loop.  The CPU oracle
needs 0.1-0.4 s per CTU, so it checks a deterministic sample; every CTU of every plane is covered by
size-independent properties: both search kernels agree, the returned cost is the true SAD at the returned
vector (recomputed by the independent batched-SAD kernel), the vector is legal and no worse than the zero
vector and random legal candidates.

Run on the MI355X box:  python -u -m pytest tests -m gpu -x -v
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _batch_properties(gpu, oracle, W, H, R, nframes, nrefs, oracle_step, seed=12345):
    """frames[nrefs + f] is searched against frames[nrefs + f - 1 - r] (the pipeline's geometry: a negative
    byte stride between references), all (f, r) planes in one launch per kernel."""
    from x265_amd import synth, MeTqPipeline
    seq = synth.make_sequence(W, H, nframes + nrefs, search_range=R, seed=seed)
    frames = dev(seq)
    sea = MeTqPipeline(frames, ctu=64, search_range=R, nrefs=nrefs, method="sea")
    sea.motion_search()
    assert sea.method == "sea"
    mv = gpu.mv_to_numpy(sea.mv)                     # [F][R][nctu]
    brute = MeTqPipeline(frames, ctu=64, search_range=R, nrefs=nrefs, method="brute")
    brute.motion_search()
    np.testing.assert_array_equal(mv, gpu.mv_to_numpy(brute.mv))
    del brute
    ncx = W // 64
    nctu = mv.shape[2]
    cx = (np.arange(nctu) % ncx) * 64
    cy = (np.arange(nctu) // ncx) * 64
    flat = frames.view(-1)
    rng = np.random.default_rng(0)
    checked = 0
    for f in range(nframes):
        for r in range(nrefs):
            m = mv[f, r]
            ci, ri = nrefs + f, nrefs + f - 1 - r
            # (a) the CPU oracle on CTUs first, first + step, ..
            first = (37 * (f * nrefs + r)) % oracle_step
            want = oracle.me_fullsearch(seq[ci], seq[ri], 64, R, first=first, step=oracle_step, count=-1, nthreads=16)
            idx = np.arange(first, nctu, oracle_step)
            np.testing.assert_array_equal(m[idx], want[idx], f"frame {f} ref {r}")
            checked += len(idx)
            # (b) legality, cost == SAD at the vector, no worse than 1 + 4 other legal candidates
            assert ((cx + m["mvx"] >= 0) & (cx + m["mvx"] + 64 <= W) & (cy + m["mvy"] >= 0) & (cy + m["mvy"] + 64 <= H)).all()
            assert (np.abs(m["mvx"]) <= R).all() and (np.abs(m["mvy"]) <= R).all()
            so = dev((ci * H * W + cy * W + cx).astype(np.int32))
            ro = dev((ri * H * W + (cy + m["mvy"]) * W + cx + m["mvx"]).astype(np.int32))
            np.testing.assert_array_equal(host(gpu.sad_blocks(64, 64, flat, W, flat, W, so, ro)).astype(np.uint32), m["sad"])
            for t in range(5):
                dx = np.zeros(nctu, np.int64) if t == 0 else rng.integers(-R, R + 1, nctu)
                dy = np.zeros(nctu, np.int64) if t == 0 else rng.integers(-R, R + 1, nctu)
                dx = np.clip(cx + dx, 0, W - 64) - cx
                dy = np.clip(cy + dy, 0, H - 64) - cy
                ro = dev((ri * H * W + (cy + dy) * W + cx + dx).astype(np.int32))
                s = host(gpu.sad_blocks(64, 64, flat, W, flat, W, so, ro)).astype(np.uint32)
                assert (m["sad"] <= s).all()
    return checked


def test_me_8k_four_references_full_size(gpu, oracle):
    """BASELINE.json configs[4] as written (the whole GOP resident): 7680x4320 (padded to 4352), 4
    reference frames, R 64, EIGHT frames per launch: 8160 CTUs x 32 planes, 33 MB planes, 9 GB of workspace -- the
    sizes at which plane / CTU / workspace indexing could overflow.  Oracle on every 499th CTU of every (frame,
    reference) plane."""
    n = _batch_properties(gpu, oracle, 7680, 4352, 64, 8, 4, 499)
    assert n >= 32 * 16


def test_me_4k_gop_segment_batch(gpu, oracle):
    """BASELINE.json configs[3] in the bench's shape: 3840x2160 (padded to 2176), R 128, one GOP segment of
    FOUR frames per launch -- the 16-wave exact kernel reusing its workspace schedule slices across CTUs and
    planes.  Oracle on every 97th CTU of every frame."""
    n = _batch_properties(gpu, oracle, 3840, 2176, 128, 4, 1, 97)
    assert n >= 4 * 21


def test_tq_1080p_chroma_plane_mv_shift(gpu, oracle):
    """The chroma leg of the closed loop at full size: a 960x544 plane (4:2:0 of 1920x1088), 16x16 TUs at
    g_aucChromaScale[32] (src/encode.cpp:596), prediction = reference chroma displaced by the LUMA vectors >> 1
    (mv_shift = 1, one vector per 32x32 chroma block), P-slice rounding -- against the oracle fed the halved vectors."""
    import hashlib
    from x265_amd import synth
    W, H, R = 1920, 1088, 64
    y, u, v = synth.make_sequence(W, H, 2, search_range=R, seed=12345, chroma=True)
    mv = gpu.me_fullsearch(dev(y[1]), dev(y[0]), 64, R)
    omv = gpu.mv_to_numpy(mv)[0, 0]
    half = omv.copy()
    half["mvx"] >>= 1
    half["mvy"] >>= 1
    qpc = gpu.chroma_qp(32)
    assert qpc == oracle.chroma_qp(32)
    for c in (u, v):
        level, sums, rec = gpu.tq_roundtrip(dev(c[1]), dev(c[0]), 16, qpc, gpu.SLICE_P, mv=mv.view(1, -1, 2), mv_ctu=32,
                                            mv_shift=1)
        ol, os_ = oracle.tq_forward(c[1], c[0], 16, qpc, 1, mv=half, mv_ctu=32, nthreads=8)
        orec = oracle.tq_inverse(ol, c[0], 16, qpc, 1, mv=half, mv_ctu=32, nthreads=8)
        np.testing.assert_array_equal(host(level)[0], ol)
        np.testing.assert_array_equal(host(sums)[0].astype(np.uint32), os_)
        assert hashlib.md5(host(rec)[0].tobytes()).hexdigest() == hashlib.md5(orec.tobytes()).hexdigest()


def test_bench_two_ranks_as_a_fresh_child(gpu):
    """`python bench.py --gpus 2` started by hand on this box: the script launches its own two ranks before anything
    touches the GPU (here they share the one device and meet over gloo), each searches and transforms its own shard of
    ONE sequence, rank 0 prints the line.  Checks the line's shape, that both shards report a result digest and that
    each rank's sample agreed with the CPU checker."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lean"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["metric"] == "ME+TQ Mpixels/s" and line["value"] > 0
    shards = line["shards"]
    assert [s["rank"] for s in shards] == [0, 1]
    assert shards[1]["first_frame"] == shards[0]["first_frame"] + shards[0]["frames"]      # consecutive runs of one sequence
    assert all(len(s["mv_md5"]) == 32 and len(s["level_rec_md5"]) == 32 for s in shards)
    assert shards[0]["mv_md5"] != shards[1]["mv_md5"]                                        # different frames, different vectors
    assert all(s["sample_parity_vs_cpu"] for s in shards)
