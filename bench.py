#!/usr/bin/env python3
"""bench.py -- ME+TQ throughput of the MI355X hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p|4k|8k4ref|tq1080p|gop1080p|intra1080p]

One STEP = one pass of the hot path over one batch of synthetic frames that is
already resident in HBM: exhaustive +-R integer-pel search for every 64x64 CTU
of every frame (x265amd_me_fullsearch_sea), then residual -> DCT -> quant -> dequant
-> IDCT -> reconstruction with the motion-compensated prediction
(x265amd_tq_roundtrip).  Default workload = BASELINE.json configs[1]:
1920x1080 (padded to 1088 rows), CTU 64, R = 64, 16 frames per step, 1 reference.

N > 1: one process per GPU.  Either the driver starts them (torchrun: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when WORLD_SIZE is unset, this script starts N children of itself BEFORE anything touches
the GPU and rank 0's line is the output.  ONE synthetic sequence of N x frames-per-step frames is sharded
with x265_amd.shard_frames: rank r searches and transforms its own contiguous run of frames (weak scaling:
the per-GPU batch is fixed; no data-path collective -- SURVEY.md section 8e).  torch.distributed provides the
barrier, the max-over-ranks time and the gather of every rank's result digest and sample-parity flag to rank 0.
With fewer GPUs than ranks (a rehearsal on a 1-GPU box) the ranks share devices and the group is gloo.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the
search) from HIP-event timing on the launch stream inside the timed region;
`cpu_baseline` times the reference (oracle/_ref) or the CPU port (oracle/) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import x265_amd  # noqa: E402
from x265_amd import synth  # noqa: E402
from x265_amd.pipeline import MeTqPipeline  # noqa: E402
from x265_amd.sharding import shard_frames  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s HBM3E
QSAD_PEAK_TPX = 152.0       # tools/ubench_valu: v_qsad_pk_u16_u8 issue rate on this chip (profiles/ubench_r1.txt)

WORKLOADS = {
    # name: (nominal W, nominal H, CTU, R, nrefs, frames per step, TU, QP)
    "1080p": (1920, 1080, 64, 64, 1, 16, 32, 32),     # BASELINE.json configs[1]
    "4k": (3840, 2160, 64, 128, 1, 4, 32, 32),        # configs[3]: one GOP segment of 4 per step
    "8k4ref": (7680, 4320, 64, 64, 4, 8, 32, 32),     # configs[4]: the whole GOP of 8 frames resident per GPU
    "tq1080p": (1920, 1080, 64, 0, 1, 16, 32, 32),    # configs[2]: transform/quant only (see --tu), no search
    "gop1080p": (1920, 1080, 64, 64, 1, 16, 32, 32),  # closed-loop P pictures, 16 GOP segments, 4:2:0
    "intra1080p": (1920, 1080, 32, 0, 0, 16, 32, 32),  # the reference's own picture type, CU 32, 16 frames per step
}


# ------------------------------------------------------------------------------------------------
# launching
# ------------------------------------------------------------------------------------------------
def dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return world, rank, local


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: N children of this script, one per GPU.  Runs before this
    process has made any GPU call (torch.cuda.device_count() makes none on this image) and starts nothing
    after one; the children are fresh interpreters.  Rank 0 inherits stdout, so its JSON line is the output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ndev = torch.cuda.device_count()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        if ndev < n:
            env.setdefault("X265AMD_DIST_BACKEND", "gloo")   # rehearsal: ranks share devices, no RCCL ring possible
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class Group:
    """The process group as the bench uses it: barrier, max over ranks, gather of small python objects."""

    def __init__(self, world, rank, backend):
        self.world, self.rank, self.backend = world, rank, backend
        self.dist = None
        if world > 1:
            import torch.distributed as dist
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()
        if torch.cuda.is_available():   # (the CPU test of the rank logic has no device to wait for)
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """every rank's object, in rank order (on every rank)"""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def setup(args):
    world, rank, local = dist_env()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    # one process per GPU (nccl = RCCL).  gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks.
    backend = os.environ.get("X265AMD_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    dev_index = local if backend == "nccl" else local % ndev
    torch.cuda.set_device(dev_index)
    x265_amd.init(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    return Group(world, rank, backend)


# ------------------------------------------------------------------------------------------------
# workload pieces
# ------------------------------------------------------------------------------------------------
def make_frames(nw, nh, ctu, rng, nframes, seed):
    """Nominal-size synthetic luma, edge-replicated to a multiple of the CTU (1080 -> 1088)."""
    seq = synth.make_sequence(nw, nh, nframes, search_range=rng, seed=seed)
    return np.stack([synth.pad_plane(f, ctu) for f in seq])


def shard_of_sequence(nw, nh, ctu, rng, nrefs, fps, world, rank, seed=12345):
    """ONE sequence of nrefs + world * fps frames (LCG seed 12345); this rank's run of `fps`
    frames plus the nrefs frames before it.  The generator is sequential in time, so a rank generates the
    prefix that ends with its own shard -- frame for frame what every other rank would generate for it."""
    total = nrefs + world * fps
    # shard_frames splits frames 1 .. n-1 of a sequence whose frame 0 is only a reference: with nrefs
    # leading reference frames, frame index = nrefs - 1 + its index
    sh = shard_frames(total - (nrefs - 1), world, rank, nrefs=1, segment=fps)
    first = sh.first + nrefs - 1
    assert sh.count == fps and first == nrefs + rank * fps
    frames = make_frames(nw, nh, ctu, rng, first + fps, seed)
    return frames[first - nrefs:], first


def me_algorithmic_bytes(w, h, nctu, nrefs):
    """every luma sample of cur and of each ref once + 8 B per CTU per ref."""
    return w * h * (1 + nrefs) + nctu * nrefs * 8


def legal_candidates(w, h, ctu, rng):
    """sum over CTUs of the candidates that keep the block inside the picture (App. A.7)"""
    n = 0
    for cy in range(0, h, ctu):
        ny = min(rng, cy) + min(rng, h - ctu - cy) + 1
        for cx in range(0, w, ctu):
            n += (min(rng, cx) + min(rng, w - ctu - cx) + 1) * ny
    return n


def evaluated_half_groups(pipe):
    """One extra, untimed pass with the statistics variant of the elimination kernels: HALF blocks of groups of four
    candidates whose SAD was evaluated (upper halves of everything scheduled + the lower halves that were finished),
    summed over the batch."""
    lib = x265_amd.api._lib()
    stats = torch.zeros((pipe.nframes * pipe.nrefs, pipe.nctu, 16), dtype=torch.int32, device="cuda")
    lib.x265amd_debug_set_me_stats(stats.data_ptr())
    try:
        pipe.motion_search()
        torch.cuda.synchronize()
    finally:
        lib.x265amd_debug_set_me_stats(None)
    s = stats.cpu().numpy().astype(np.int64)
    # half blocks: every scheduled group's upper half ([0] first batch, [1] second pass) + the lower halves that were
    # finished ([15], [14]); a second pass of at most 64 groups goes whole and is counted in both
    return int(s[..., 0].sum() + s[..., 1].sum() + s[..., 14].sum() + s[..., 15].sum())


def kernel_split(pipe, reps):
    """per-kernel ms of the search stage: library-side HIP events on the launch stream"""
    import ctypes
    lib = x265_amd.api._lib()
    lib.x265amd_debug_me_timing(1)
    acc = np.zeros(4)
    buf = (ctypes.c_float * 4)()
    try:
        for _ in range(reps):
            pipe.motion_search()
            if lib.x265amd_debug_me_last_ms(buf) != 0:
                return None
            acc += np.array(buf[:])
    finally:
        lib.x265amd_debug_me_timing(0)
    return {k: round(float(v) / reps, 4) for k, v in zip(["box16_pairs", "_", "sea_bounds_kernel", "sea_exact_kernel"], acc)
            if k != "_"}


def other_content(w, h, ctu, rng, nrefs, nominal_px, which):
    """The search stage on other content than the headline's, 4 frames per launch, both entry points; identical output
    required.  which = "worst": white noise (bounds useless) and a flat picture (every candidate ties);
    "mid": 1/f-like texture in which every CTU moves with its own vector (synth.make_textured_sequence)."""
    out = {}
    g = np.random.default_rng(0)
    nf = 4
    if which == "worst":
        cases = (("noise", g.integers(0, 256, (nf + nrefs, h, w), dtype=np.uint8)),
                 ("flat", np.full((nf + nrefs, h, w), 77, np.uint8)))
    else:
        cases = (("textured", synth.make_textured_sequence(w, h, nf + nrefs, search_range=rng, ctu=ctu)),)
    for name, frames in cases:
        fr = torch.from_numpy(frames).cuda()
        res = {}
        # both entry points alternately, after a common warm-up: measured one after the other, the first pays for the
        # clocks that ramp up after the idle stretch in which the content was made (the every-candidate kernel's own
        # time falls from 1.09 to 1.06 ms over its first ten launches)
        methods = ("sea", "brute")
        pipes = {m: MeTqPipeline(fr, ctu=ctu, search_range=rng, nrefs=nrefs, method=m) for m in methods}
        for _ in range(6):
            for m in methods:
                pipes[m].motion_search()
        torch.cuda.synchronize()
        times = {m: [] for m in methods}
        for _ in range(4):
            for m in methods:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pipes[m].motion_search()
                e1.record()
                torch.cuda.synchronize()
                times[m].append(e0.elapsed_time(e1))
        for m in methods:
            ms = float(np.mean(times[m]))
            res[f"{m}_ms"] = round(ms, 4)
            res[f"{m}_search_Mpixels_per_s"] = round(nominal_px * nf / (ms * 1e-3) / 1e6, 1)
        res["identical_output"] = bool(torch.equal(pipes["sea"].mv, pipes["brute"].mv))
        del pipes
        out[name] = res
    if which == "worst":
        out["note"] = ("search stage only, 4 frames per launch, the two entry points timed alternately; sea = x265amd_me_fullsearch_sea (the measured path), brute = "
                       "x265amd_me_fullsearch (every candidate's SAD).  Noise: the bounds eliminate nothing, the blocks go to the "
                       "every-candidate kernel behind the bounds pass (me_list_qsad); flat: every candidate ties, the search ends "
                       "after the first batch (ties go to the lower raster index).  The headline's rate does not apply here")
    else:
        out["note"] = ("search stage only, 4 frames per launch: 1/f-like texture, every CTU with its own vector in [-R/2, R/2]^2 "
                       "(synth.make_textured_sequence) -- between the headline's smooth content and white noise")
    return out


def cpu_baseline(frames_np, ctu, rng, nrefs, tu, qp, gpu_mv, gpu_level, gpu_rec, nominal_px_per_frame):
    """The reference (oracle/_ref, else the port) on the host cores: ME on a deterministic sample of CTUs,
    TQ on one whole frame; all cores and one core.  Also the in-run parity check of the GPU result."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    threads = min(16, os.cpu_count() or 1)
    orc = pyoracle.Oracle()
    kind = "port"
    me_lib = orc
    ref = None
    if pyoracle.have_reference():
        me_lib = ref = pyoracle.Reference()
        kind = "reference"
    h, w = frames_np.shape[1:]
    nctu = (h // ctu) * (w // ctu)
    cur, prev = frames_np[nrefs], frames_np[nrefs - 1]
    # one core first, on a few CTUs spread over the frame: it calibrates the sample (the compiled reference's
    # xSad64xN vectorises to ~10 ms per CTU on the box's cores, ~95 ms here)
    n1 = max(2, min(nctu, 16))
    step1 = max(nctu // n1, 1)
    t0 = time.perf_counter()
    me_lib.me_fullsearch(cur, prev, ctu, rng, first=step1 // 2, step=step1, count=n1, nthreads=1)
    t_me1 = (time.perf_counter() - t0) / n1
    # all cores: ~15 core-seconds of CTUs, the whole frame if that is less
    count = int(max(threads * 2, min(nctu, 15.0 / t_me1)))
    step = max(nctu // count, 1)
    first = 0 if count == nctu else step // 2
    t0 = time.perf_counter()
    mv = me_lib.me_fullsearch(cur, prev, ctu, rng, first=first, step=step, count=count, nthreads=threads)
    t_me = time.perf_counter() - t0
    idx = np.arange(first, nctu, step)[:count]
    parity = bool((mv[idx] == gpu_mv[idx]).all())

    # TQ on the whole frame with the GPU's vectors (identical to the CPU's where sampled).  The reference's
    # xSubDct / xQuant / xDeQuant / xIDctAdd are single-threaded code: all cores = one CTU row of TUs per task.
    def tq_ref_band(y0):
        y1 = min(y0 + ctu, h)
        ncx = w // ctu
        return ref.tq_frame(cur[y0:y1], prev[y0:], tu, qp, 1, mv=gpu_mv[(y0 // ctu) * ncx:], mv_ctu=ctu)

    if ref is not None:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            bands = list(ex.map(tq_ref_band, range(0, h, ctu)))
        t_tq = time.perf_counter() - t0
        lvl = np.concatenate([b[0] for b in bands])
        rec = np.concatenate([b[2] for b in bands])
        t0 = time.perf_counter()
        ref.tq_frame(cur, prev, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu)
        t_tq1 = time.perf_counter() - t0
        tq_how = "oracle/_ref xSubDct + xQuant + xDeQuant + xIDctAdd, one CTU row per task"
    else:
        t0 = time.perf_counter()
        lvl, _ = orc.tq_forward(cur, prev, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=threads)
        rec = orc.tq_inverse(lvl, prev, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=threads)
        t_tq = time.perf_counter() - t0
        t0 = time.perf_counter()
        l1, _ = orc.tq_forward(cur, prev, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=1)
        orc.tq_inverse(l1, prev, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=1)
        t_tq1 = time.perf_counter() - t0
        tq_how = "oracle port"
    parity = parity and bool((lvl == gpu_level).all()) and bool((rec == gpu_rec).all())
    # the CPU searches every reference (timed on reference 0, the previous frame: the same work per reference)
    sec_per_frame = t_me / len(idx) * nctu * nrefs + t_tq
    sec_per_frame_1 = t_me1 * nctu * nrefs + t_tq1
    return {
        "value": round(nominal_px_per_frame / sec_per_frame / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": kind,
        "one_core_value": round(nominal_px_per_frame / sec_per_frame_1 / 1e6, 4),
        "sample": f"ME: {len(idx)} of {nctu} CTUs of one frame ({'all' if step == 1 else f'every {step}th'}) in {t_me:.2f} s wall "
                  f"on {threads} threads ({n1} CTUs on one core: {t_me1 * 1e3:.1f} ms per CTU), scaled to the frame{f' and its {nrefs} references' if nrefs > 1 else ''} "
                  f"({'oracle/_ref xSadN loop' if kind == 'reference' else 'oracle port'}); TQ fwd+inv: one whole frame in "
                  f"{t_tq:.3f} s on {threads} threads, {t_tq1:.3f} s on one core ({tq_how})",
        "parity_with_gpu_on_sample": parity,
    }


def sample_parity(frames_np, nrefs, ctu, rng, mv_np, first_frame, nsample=8):
    """A few CTUs of this rank's first frame against the CPU checker (reference loop if built, else the port):
    what rank 0 gathers from every rank of a multi-GPU run."""
    from oracle import pyoracle
    lib = pyoracle.Reference() if pyoracle.have_reference() else pyoracle.Oracle()
    h, w = frames_np.shape[1:]
    nctu = (h // ctu) * (w // ctu)
    step = max(nctu // nsample, 1)
    first = (first_frame * 7) % step
    want = lib.me_fullsearch(frames_np[nrefs], frames_np[nrefs - 1], ctu, rng, first=first, step=step, count=nsample,
                             nthreads=min(8, os.cpu_count() or 1))
    idx = np.arange(first, nctu, step)[:nsample]
    return bool((want[idx] == mv_np[0, 0][idx]).all())


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def timed(grp, args, step_fn):
    for _ in range(args.warmup):
        step_fn()
    grp.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_fn(k)
    grp.barrier()
    return grp.max_over_ranks(time.perf_counter() - t0)


def tq_only(args, grp, nw, nh, ctu, fps, tu, qp):
    """BASELINE configs[2]: the intra-style transform/quant path alone -- residual against the co-located
    block of the previous frame, forward transform + quant + dequant + inverse + reconstruction
    (x265amd_tq_roundtrip), SLICE_I rounding, no search."""
    world, rank = grp.world, grp.rank
    frames_np, first = shard_of_sequence(nw, nh, ctu, 16, 1, fps, world, rank)
    h, w = frames_np.shape[1:]
    frames = torch.from_numpy(frames_np).cuda()
    cur, pred = frames[1:], frames[:-1]
    level = torch.empty((fps, h, w), dtype=torch.int16, device="cuda")
    sums = torch.empty((fps, (h // tu) * (w // tu)), dtype=torch.int32, device="cuda")
    rec = torch.empty((fps, h, w), dtype=torch.uint8, device="cuda")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(k=None):
        if k is not None:
            ev[k][0].record()
        x265_amd.tq_roundtrip(cur, pred, tu, qp, x265_amd.SLICE_I, out=(level, sums, rec))
        if k is not None:
            ev[k][1].record()

    elapsed = timed(grp, args, step)
    digest = hashlib.md5(level.cpu().numpy().tobytes() + rec.cpu().numpy().tobytes()).hexdigest()
    digests = grp.gather({"rank": rank, "first_frame": first, "frames": fps, "level_rec_md5": digest})
    if rank == 0:
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        tq_bytes = 5 * w * h * fps
        out = {
            "metric": "TQ Mpixels/s", "value": round(nw * nh * fps * world * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 8-bit luma (padded {w}x{h}), {tu}x{tu} residual + DCT + quant + dequant + "
                                   f"IDCT + recon, QP{qp}, I-slice rounding, prediction = co-located block of the previous "
                                   f"frame; {fps} frames per step per GPU, one sequence sharded across GPUs", "name": args.workload,
                       "tu": tu},
            "roofline": {"kernel": f"tq roundtrip kernel, {tu}x{tu}", "bound": "hbm",
                         "achieved": round(tq_bytes / (ms * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tq_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic(args.workload if tu == 32 else f"{args.workload}_tu{tu}", "tq_bytes_per_launch"),
                         "algorithmic_bytes_per_launch": tq_bytes, "avg_launch_ms": round(ms, 4)},
            "shards": digests,
        }
        if world == 1 and not args.lean:
            from oracle import pyoracle
            orc = pyoracle.Oracle()
            threads = min(16, os.cpu_count() or 1)
            t0 = time.perf_counter()
            lvl, _ = orc.tq_forward(frames_np[1], frames_np[0], tu, qp, 0, nthreads=threads)
            orec = orc.tq_inverse(lvl, frames_np[0], tu, qp, 0, nthreads=threads)
            dt = time.perf_counter() - t0
            base = {"value": round(nw * nh / dt / 1e6, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                    "sample": f"one whole frame, forward + inverse, in {dt:.3f} s (oracle port, OpenMP)",
                    "parity_with_gpu_on_sample": bool((lvl == level[0].cpu().numpy()).all()
                                                      and (orec == rec[0].cpu().numpy()).all())}
            if pyoracle.have_reference():
                t0 = time.perf_counter()
                rl, _, rr = pyoracle.Reference().tq_frame(frames_np[1], frames_np[0], tu, qp, 0)
                dt1 = time.perf_counter() - t0
                base["one_core_value"] = round(nw * nh / dt1 / 1e6, 2)
                base["sample"] += f"; one core: the reference's own xSubDct/xQuant/xDeQuant/xIDctAdd (oracle/_ref) in {dt1:.3f} s"
                base["parity_with_gpu_on_sample"] = base["parity_with_gpu_on_sample"] and bool(
                    (rl == level[0].cpu().numpy()).all() and (rr == rec[0].cpu().numpy()).all())
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    grp.close()


def closed_loop(args, grp, nw, nh, rng, nseg, qp):
    """Closed-loop P pictures (x265_amd.ClosedLoopGop): `nseg` independent GOP segments advance one picture per
    step -- search against the previous RECONSTRUCTION, luma 32x32 + chroma 16x16 transform / quant /
    reconstruction, reference rotation.  Pixels counted: luma of the nominal picture, like the main metric."""
    from x265_amd import ClosedLoopGop
    world, rank = grp.world, grp.rank
    T = 4
    w, h = nw, (nh + 63) // 64 * 64
    segs = []
    for s_ in range(nseg):   # GOP segments are the shard unit: segment id = rank * nseg + s_
        y, u, v = synth.make_sequence(w, h, T, search_range=rng, seed=12345 + rank * nseg + s_, chroma=True)
        segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
    yuv = torch.from_numpy(np.stack(segs)).cuda()          # [S][T][W*H*3/2]
    enc = ClosedLoopGop(w, h, nseg, search_range=rng, qp=qp)
    enc.set_reference(yuv[:, 0].contiguous())
    # the pictures go back and forth, 1 .. T-1, T-2 .. 0, 1 ..: every step codes a picture next in time to its reference,
    # as in a real segment (1 .. T-1 over and over made every (T-1)-th step a scene cut, T-1 pictures back, and the
    # reconstructions after it noisier: tools/gop_steps.py)
    pics = [yuv[:, t].contiguous() for t in list(range(1, T)) + list(range(T - 2, -1, -1))]
    k = [0]

    def step(_k=None):
        enc.encode_p(pics[k[0] % len(pics)])
        k[0] += 1

    elapsed = timed(grp, args, step)
    if rank == 0:
        print(json.dumps({
            "metric": "closed-loop ME+TQ Mpixels/s", "value": round(nw * nh * nseg * world * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 4:2:0 (padded {w}x{h}), closed loop: {nseg} GOP segments per GPU advance one P "
                                   f"picture per step; +-{rng} search on the previous reconstruction, luma 32x32 and chroma "
                                   f"16x16 TQ + reconstruction QP{qp}, reference rotation; pictures in time order back and forth", "name": args.workload,
                       "me_method": enc.method},
        }), flush=True)
    grp.close()


def intra_frames(args, grp, nw, nh, cu, fps, qp):
    """The pixel path of the reference's intra encoder (x265amd_intra_encode_frames): `fps` independent frames per
    step share every anti-diagonal step.  4:2:0; pixels counted: luma of the nominal picture, like the main metric."""
    world, rank = grp.world, grp.rank
    w, h = nw, (nh + cu - 1) // cu * cu
    fsz = w * h * 3 // 2
    T = 8
    probe = synth.make_intra_probe(w, h, T)              # ONE sequence; this rank's frames = its shard of it
    idx = [(rank * fps + k) % T for k in range(fps)]
    yuv = torch.from_numpy(np.ascontiguousarray(probe[idx])).cuda()
    out = x265_amd.intra_encode_frames(yuv, w, h, cu, qp)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def step(_k=None):
        x265_amd.intra_encode_frames(yuv, w, h, cu, qp, out=out)

    elapsed = timed(grp, args, step)
    ev[0].record()
    for _ in range(5):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    dev_ms = ev[0].elapsed_time(ev[1]) / 5
    cpu = None
    if rank == 0 and not args.lean:
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        t0 = time.perf_counter()
        orec, my, mc, lv = orc.intra_encode_frame(probe[idx[0]], w, h, cu, qp, want_levels=True)
        dt = time.perf_counter() - t0
        same = (np.array_equal(out["rec"][0].cpu().numpy(), orec) and np.array_equal(out["level"][0].cpu().numpy(), lv) and
                np.array_equal(out["mode_y"][0].cpu().numpy(), my) and np.array_equal(out["mode_c"][0].cpu().numpy(), mc))
        if not same:
            raise SystemExit("intra1080p: GPU frame 0 differs from the oracle's restatement")
        cpu = {"value": round(nw * nh / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": f"one {w}x{h} frame through orc_intra_encode_frame ({dt * 1e3:.0f} ms); it is also the in-run parity "
                         "check of the GPU's frame 0 (reconstruction, levels, modes: identical)"}
    if rank == 0:
        nsteps = w // cu + 2 * (h // cu - 1)
        alg = fps * fsz * (1 + 1 + 2)   # source in, reconstruction out, Int16 levels out (4:2:0: 1.5 samples per pixel)
        print(json.dumps({
            "metric": "intra pixel-path Mpixels/s", "value": round(nw * nh * fps * world * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 4:2:0 (padded {w}x{h}) intra pictures, CU = PU = TU = {cu}, QP{qp}, SLICE_I: "
                                   f"{fps} independent frames per step per GPU through x265amd_intra_encode_frames "
                                   f"({nsteps} anti-diagonal steps, one launch each: prediction + decisions + transform stage of the three planes)",
                       "name": args.workload, "frames_per_step_per_gpu": fps},
            "roofline": {"kernel": "the whole step (intra_wave_tq_kernel per anti-diagonal: prediction + transform stage)", "bound": "hbm",
                         "achieved": round(alg / (dev_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": intra_traffic(args.workload, nsteps),
                         "note": "algorithmic bytes = source + reconstruction + Int16 levels; the step is a chain of "
                                 f"{nsteps} dependent launches of at most {fps} x {min(h // cu, (w // cu + 1) // 2)} workgroups, "
                                 "bound by that chain's latency, not by HBM or the VALU (DESIGN.md 4.6)"},
            "cpu_baseline": cpu,
        }), flush=True)
    grp.close()


def intra_traffic(key, nsteps):
    """bytes per step from the committed PMC passes: every dispatch of the step's kernels (per-dispatch averages x
    dispatches) over the number of steps the counted run made (= dispatches of the prediction kernel / anti-diagonals)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ks = json.load(f)[key]["kernels"]
        pred = [k for k in ks if k.startswith(("intra_wave_kernel", "intra_wave_tq_kernel"))]
        steps = sum(ks[k]["dispatches"] for k in pred) / nsteps
        tot = sum(v["dispatches"] * (v["fetch_bytes_x2"] + v["write_bytes"]) for k, v in ks.items()
                  if k.startswith(("intra_wave_kernel", "intra_wave_tq_kernel", "tq_mfma", "tq_kernel")))
        return round(tot / steps)
    except Exception:
        return None


def pmc_traffic(key, field):
    """HBM-side bytes per launch from the PMC passes committed under profiles/ (tools/run_pmc.sh; separate
    --pmc runs, gfx950 corrections): counters cannot be collected inside this run."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(key, {}).get(field)
    except Exception:
        return None


def search_and_tq(args, grp, nw, nh, ctu, rng, nrefs, fps, tu, qp):
    world, rank = grp.world, grp.rank
    frames_np, first = shard_of_sequence(nw, nh, ctu, rng, nrefs, fps, world, rank)
    h, w = frames_np.shape[1:]
    frames = torch.from_numpy(frames_np).cuda()
    pipe = MeTqPipeline(frames, ctu=ctu, search_range=rng, nrefs=nrefs, tu=tu, qp=qp, method=args.me_method)
    nctu = pipe.nctu
    # HIP events split the step into its search and transform stages live, inside the timed region -- on every
    # fourth step only: a timed event is a barrier packet on the queue, three of them cost ~5 % of a 0.7 ms step
    ev = {k: tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for k in range(0, args.steps, 4)}

    def step(k=None):
        if k not in ev:
            return pipe.step()
        ev[k][0].record()
        pipe.motion_search()
        ev[k][1].record()
        pipe.transform_quant()
        ev[k][2].record()

    elapsed = timed(grp, args, step)
    me_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev.values()]))
    tq_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev.values()]))

    # every rank: digest of its results + a small sample of its own first frame against the CPU checker
    torch.cuda.synchronize()
    mv_np = x265_amd.mv_to_numpy(pipe.mv)
    shard = {"rank": rank, "first_frame": first, "frames": fps,
             "mv_md5": hashlib.md5(mv_np.tobytes()).hexdigest(),
             "level_rec_md5": hashlib.md5(pipe.level.cpu().numpy().tobytes() + pipe.rec.cpu().numpy().tobytes()).hexdigest()}
    if world > 1:
        shard["sample_parity_vs_cpu"] = sample_parity(frames_np, nrefs, ctu, rng, mv_np, first)
    shards = grp.gather(shard)

    if rank == 0:
        kernels_ms = kernel_split(pipe, min(args.steps, 10)) if pipe.method == "sea" else None
        px_step = nw * nh * fps                      # nominal pixels (1920x1080), SURVEY.md 8d
        value = px_step * world * args.steps / elapsed / 1e6
        me_bytes = me_algorithmic_bytes(w, h, nctu, nrefs) * fps
        achieved = me_bytes / (me_ms * 1e-3) / 1e9
        tq_bytes = 5 * w * h * fps                   # src + pred + 2 B level + rec
        exhaustive = legal_candidates(w, h, ctu, rng) * ctu * ctu * nrefs * fps
        valu = {"peak_Tpx_sad_per_s": QSAD_PEAK_TPX,
                "peak_source": "tools/ubench_valu: v_qsad_pk_u16_u8 issue rate on this chip (profiles/ubench_r1.txt)",
                "exhaustive_pixel_sads_per_launch": exhaustive,
                "exhaustive_equivalent_Tpx_sad_per_s": round(exhaustive / (me_ms * 1e-3) / 1e12, 3)}
        if pipe.method == "sea" and not args.lean:
            halves = evaluated_half_groups(pipe)
            ev_sads = halves * 4 * ctu * ctu // 2    # a group = four candidates, half a block = ctu x ctu / 2 pixels each
            valu.update({
                "evaluated_pixel_sads_per_launch": ev_sads,
                "evaluated_fraction_of_exhaustive": round(ev_sads / exhaustive, 5),
                "achieved_Tpx_sad_per_s": round(ev_sads / (me_ms * 1e-3) / 1e12, 3),
                "frac": round(ev_sads / (me_ms * 1e-3) / 1e12 / QSAD_PEAK_TPX, 4),
                "frac_exact_kernel_alone": round(ev_sads / (kernels_ms["sea_exact_kernel"] * 1e-3) / 1e12 / QSAD_PEAK_TPX, 4)
                if kernels_ms else None,
                "note": "achieved = pixel-SADs the elimination search actually evaluated (statistics variant of the "
                        "kernels, one extra untimed pass; counted in half blocks: a group dropped after its upper half "
                        "counts half) / search-stage time: the stage's real qsad issue rate against "
                        "the measured issue peak.  exhaustive_equivalent = what an every-candidate search would have "
                        "had to evaluate / the same time (an effective rate, not a utilisation)"})
        elif pipe.method == "brute":
            valu.update({"achieved_Tpx_sad_per_s": valu["exhaustive_equivalent_Tpx_sad_per_s"],
                         "frac": round(valu["exhaustive_equivalent_Tpx_sad_per_s"] / QSAD_PEAK_TPX, 4)})
        out = {
            "metric": "ME+TQ Mpixels/s",
            "value": round(value, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 8-bit luma (padded {w}x{h}), {ctu}x{ctu} CTU, +-{rng} integer-pel "
                                   f"full search, {nrefs} ref(s), then {tu}x{tu} DCT+quant+recon QP{qp}; "
                                   f"{fps} frames per step per GPU, one sequence of {world * fps} frames sharded across "
                                   f"GPUs (x265_amd.shard_frames), no collective",
                       "name": args.workload, "frames_per_step_per_gpu": fps, "me_method": pipe.method,
                       "dist_backend": grp.backend if world > 1 else None},
            "roofline": {"kernel": "search stage = box16_pairs + sea_bounds_kernel + sea_exact_kernel (dominant)"
                                   if pipe.method == "sea" else "me_fullsearch_qsad",
                         "bound": "valu",
                         "bound_note": "integer VALU (v_qsad_pk_u16_u8 issue) binds this stage, see `valu`; achieved / peak / frac "
                                       "below are the HBM figures BASELINE.json's metric asks for (algorithmic bytes / stage time)",
                         "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic(args.workload, "me_bytes_per_launch"),
                         "traffic_source": "committed PMC pass, not measured in this run: profiles/pmc_traffic.json (rocprofv3 --pmc "
                                           "FETCH_SIZE / WRITE_SIZE passes of this command, tools/run_pmc.sh)",
                         "algorithmic_bytes_per_launch": me_bytes,
                         "avg_launch_ms": round(me_ms, 4), "kernels_ms": kernels_ms,
                         "note": "avg_launch_ms comes from the event-instrumented steps (every fourth), which run ~2 % slower than "
                                 "the plain ones that ms_per_step averages over: every frac in this line is understated by "
                                 "about that much.  avg_launch_ms = the whole search stage per step (sum of its kernels; compare the "
                                 "sum of their averages in profiles/*kernel_stats.csv).  Exhaustive SAD search is "
                                 "integer-VALU bound, not HBM bound (SURVEY.md 8d): `valu` is its roofline.  traffic "
                                 "counts fabric requests, Infinity-Cache hits included (window overlap between "
                                 "neighbouring CTUs, box-sum workspace)"},
            "valu": valu,
            "roofline_tq": {"kernel": "tq_roundtrip", "bound": "hbm", "achieved": round(tq_bytes / (tq_ms * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(tq_bytes / (tq_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "traffic": pmc_traffic(args.workload, "tq_bytes_per_launch"),
                            "algorithmic_bytes_per_launch": tq_bytes, "avg_launch_ms": round(tq_ms, 4)},
            "shards": shards,
        }
        if world == 1 and not args.lean:
            gmv = mv_np[0, 0]
            out["cpu_baseline"] = cpu_baseline(frames_np, ctu, rng, nrefs, tu, qp, gmv, pipe.level[0].cpu().numpy(),
                                               pipe.rec[0].cpu().numpy(), nw * nh)
            if args.workload == "1080p":
                out["mid_case"] = other_content(w, h, ctu, rng, nrefs, nw * nh, "mid")
                out["worst_case"] = other_content(w, h, ctu, rng, nrefs, nw * nh, "worst")
        print(json.dumps(out), flush=True)
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip needs some 30 launches after idle to reach its steady clocks (0.766 ms per step after 3
    # warm-up steps, 0.72 after 40: profiles/r2_warmup.txt); 100 timed steps are 0.1 s
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--me-method", default="sea", choices=["sea", "brute"],
                    help="sea: exact successive elimination (default); brute: every candidate's SAD. Same output.")
    ap.add_argument("--tu", type=int, default=None, choices=[4, 8, 16, 32], help="transform size (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", "--lean", dest="lean", action="store_true",
                    help="timed region only: no CPU baseline, no statistics pass, no worst-case sub-line (profiler runs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))     # nothing in this process has touched the GPU
    grp = setup(args)
    nw, nh, ctu, rng, nrefs, fps, tu, qp = WORKLOADS[args.workload]
    tu = args.tu or tu
    if args.workload == "tq1080p":
        return tq_only(args, grp, nw, nh, ctu, fps, tu, qp)
    if args.workload == "gop1080p":
        return closed_loop(args, grp, nw, nh, rng, fps, qp)
    if args.workload == "intra1080p":
        return intra_frames(args, grp, nw, nh, ctu, fps, qp)
    return search_and_tq(args, grp, nw, nh, ctu, rng, nrefs, fps, tu, qp)


if __name__ == "__main__":
    main()
