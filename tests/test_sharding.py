"""Frame sharding across ranks (no data-path collective), covered on CPU
with a real 2-process gloo group."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_frames():
    from x265_amd.sharding import shard_frames
    for nframes in (1, 2, 9, 17, 33, 100):
        for world in (1, 2, 3, 4, 8):
            for seg in (1, 4):
                for nrefs in (1, 4):
                    owned = []
                    for r in range(world):
                        s = shard_frames(nframes, world, r, nrefs=nrefs, segment=seg)
                        owned += list(range(s.first, s.first + s.count))
                        assert s.ref_first == max(s.first - nrefs, 0)
                        if s.count and r < world - 1:
                            assert s.count % seg == 0 or s.first + s.count == nframes
                    assert owned == list(range(1, nframes)), (nframes, world, seg)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from x265_amd import synth
    from x265_amd.sharding import shard_frames
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank derives the same sequence and encodes only its own frames (CPU oracle
        # stands in for the kernels here: this test is about WHO does WHAT, not the arithmetic)
        seq = synth.make_sequence(128, 64, 7, search_range=8, seed=3)
        sh = shard_frames(len(seq), world, rank)
        orc = pyoracle.Oracle()
        mine = np.zeros((len(seq), 2), pyoracle.MV_DTYPE)
        for f in range(sh.first, sh.first + sh.count):
            mine[f] = orc.me_fullsearch(seq[f], seq[f - 1], 64, 8)
        # gather the small results (motion vectors) -- the only cross-rank traffic of the design
        t = torch.from_numpy(mine.view(np.int32).reshape(len(seq), 4).copy())
        dist.all_reduce(t)  # disjoint ownership => sum == concatenation
        dist.barrier()
        if rank == 0:
            q.put(t.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_run_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from x265_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seq = synth.make_sequence(128, 64, 7, search_range=8, seed=3)
    orc = pyoracle.Oracle()
    want = np.zeros((len(seq), 2), pyoracle.MV_DTYPE)
    for f in range(1, len(seq)):
        want[f] = orc.me_fullsearch(seq[f], seq[f - 1], 64, 8)
    np.testing.assert_array_equal(got, want.view(np.int32).reshape(len(seq), 4))


# ---- bench.py's multi-rank logic (self-launch, one sharded sequence, gather to rank 0) ----
def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import hashlib
    import torch.distributed as dist
    import bench
    from oracle import pyoracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = bench.Group(world, rank, "gloo")
        nrefs, fps = 2, 3
        frames, first = bench.shard_of_sequence(128, 64, 64, 8, nrefs, fps, world, rank)
        assert frames.shape[0] == nrefs + fps and first == nrefs + rank * fps
        orc = pyoracle.Oracle()     # stands in for the kernels: this test is about who does what
        mv = np.stack([orc.me_fullsearch(frames[nrefs + f], frames[nrefs + f - 1], 64, 8) for f in range(fps)])
        grp.barrier()
        t = grp.max_over_ranks(float(rank + 1))
        shards = grp.gather({"rank": rank, "first_frame": first, "mv_md5": hashlib.md5(mv.tobytes()).hexdigest(),
                             "frames_md5": hashlib.md5(frames[nrefs:].tobytes()).hexdigest()})
        if rank == 0:
            q.put((t, shards))
    finally:
        dist.destroy_process_group()


def test_bench_rank_logic_two_ranks_gloo():
    """Two ranks shard ONE sequence: each rank's frames and vectors equal the corresponding frames of the
    unsharded sequence, rank 0 holds every rank's digest, the clock is the max over ranks."""
    import hashlib
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import bench
    from oracle import pyoracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, shards = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0 and [d["rank"] for d in shards] == [0, 1]
    nrefs, fps = 2, 3
    whole = bench.make_frames(128, 64, 64, 8, nrefs + 2 * fps, 12345)      # what ONE process would encode
    orc = pyoracle.Oracle()
    for r, d in enumerate(shards):
        first = nrefs + r * fps
        assert d["first_frame"] == first
        assert d["frames_md5"] == hashlib.md5(whole[first:first + fps].tobytes()).hexdigest()
        mv = np.stack([orc.me_fullsearch(whole[first + f], whole[first + f - 1], 64, 8) for f in range(fps)])
        assert d["mv_md5"] == hashlib.md5(mv.tobytes()).hexdigest()
    # world 1 takes the first shard of the same sequence: the N = 1 workload is a prefix of the N > 1 one
    one, first = bench.shard_of_sequence(128, 64, 64, 8, nrefs, fps, 1, 0)
    assert first == nrefs and (one == whole[:nrefs + fps]).all()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment launches two children itself; on a
    machine without a GPU each of them stops with the no-GPU message and status 2 -- never a CPU fallback."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout, r.stderr[-2000:]
    else:
        assert r.returncode == 2 and r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]
