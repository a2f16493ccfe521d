#!/usr/bin/env python3
"""bench.py -- ME+TQ throughput of the MI355X hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p|4k|8k4ref|tq1080p]

One STEP = one pass of the hot path over one batch of synthetic frames that is
already resident in HBM: exhaustive +-R integer-pel search for every 64x64 CTU
of every frame (x265amd_me_fullsearch), then residual -> DCT -> quant -> dequant
-> IDCT -> reconstruction with the motion-compensated prediction
(x265amd_tq_roundtrip).  Default workload = BASELINE.json configs[1]:
1920x1080 (padded to 1088 rows), CTU 64, R = 64, 16 frames per step, 1 reference.

N > 1: one process per GPU (torchrun); every rank runs the same-sized batch of its
own frames (weak scaling, no data-path collective -- SURVEY.md section 8e);
torch.distributed is used for the barrier and the max-over-ranks time only.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the
search) from HIP-event timing on the launch stream inside the timed region;
`cpu_baseline` times the reference (oracle/_ref) or the CPU port (oracle/) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import x265_amd  # noqa: E402
from x265_amd import synth  # noqa: E402
from x265_amd.pipeline import MeTqPipeline  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E

WORKLOADS = {
    # name: (nominal W, nominal H, CTU, R, nrefs, frames per step, TU, QP)
    "1080p": (1920, 1080, 64, 64, 1, 16, 32, 32),     # BASELINE.json configs[1]
    "4k": (3840, 2160, 64, 128, 1, 4, 32, 32),        # configs[3]: one GOP segment of 4 per step
    "8k4ref": (7680, 4320, 64, 64, 4, 2, 32, 32),     # configs[4]
    "tq1080p": (1920, 1080, 64, 0, 1, 16, 32, 32),    # configs[2]: transform/quant only (see --tu), no search
    "gop1080p": (1920, 1080, 64, 64, 1, 16, 32, 32),  # closed-loop P pictures, 16 GOP segments, 4:2:0
}


def dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return world, rank, local


def make_frames(nw, nh, ctu, rng, nframes, seed):
    """Nominal-size synthetic luma, edge-replicated to a multiple of the CTU (1080 -> 1088)."""
    seq = synth.make_sequence(nw, nh, nframes, search_range=rng, seed=seed)
    return np.stack([synth.pad_plane(f, ctu) for f in seq])


def me_algorithmic_bytes(w, h, nctu, nrefs):
    """every luma sample of cur and of each ref once + 8 B per CTU per ref."""
    return w * h * (1 + nrefs) + nctu * nrefs * 8


def cpu_baseline(frames_np, ctu, rng, nrefs, tu, qp, gpu_mv, gpu_level, gpu_rec, nominal_px_per_frame):
    """Reference (or port) ME on a deterministic sample of CTUs + TQ on one whole frame, on the
    host cores; also the in-run parity check of the GPU result on that sample."""
    from oracle import pyoracle
    threads = min(16, os.cpu_count() or 1)
    orc = pyoracle.Oracle()
    kind = "port"
    me_lib = orc
    if pyoracle.have_reference():
        me_lib = pyoracle.Reference()
        kind = "reference"
    h, w = frames_np.shape[1:]
    nctu = (h // ctu) * (w // ctu)
    cur, ref = frames_np[nrefs], frames_np[nrefs - 1]
    # size the sample for ~10-30 core-seconds of CPU work: one CTU costs ~ (2R+1)^2 * ctu^2 / 0.7e9 s per core
    per_ctu = (2 * rng + 1) ** 2 * ctu * ctu / 0.7e9
    count = int(max(threads * 2, min(nctu, 30.0 / per_ctu)))
    step = max(nctu // count, 1)
    t0 = time.perf_counter()
    mv = me_lib.me_fullsearch(cur, ref, ctu, rng, first=step // 2, step=step, count=count, nthreads=threads)
    t_me = time.perf_counter() - t0
    idx = np.arange(step // 2, nctu, step)[:count]
    parity = bool((mv[idx] == gpu_mv[idx]).all())
    # TQ on the whole frame with the GPU's vectors (identical to the CPU's where sampled)
    t0 = time.perf_counter()
    lvl, sums = orc.tq_forward(cur, ref, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=threads)
    rec = orc.tq_inverse(lvl, ref, tu, qp, 1, mv=gpu_mv, mv_ctu=ctu, nthreads=threads)
    t_tq = time.perf_counter() - t0
    parity = parity and bool((lvl == gpu_level).all()) and bool((rec == gpu_rec).all())
    sec_per_frame = t_me / len(idx) * nctu + t_tq
    return {
        "value": round(nominal_px_per_frame / sec_per_frame / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": kind,
        "sample": f"ME: {len(idx)} of {nctu} CTUs of one frame (every {step}th) in {t_me:.2f} s wall, scaled to the "
                  f"frame; TQ fwd+inv: one whole frame in {t_tq:.3f} s ({'oracle/_ref xSadN loop' if kind == 'reference' else 'oracle port'} for ME, oracle port for TQ)",
        "parity_with_gpu_on_sample": parity,
    }


def tq_only(args, world, rank, backend, nw, nh, ctu, fps, tu, qp):
    """BASELINE configs[2]: the intra-style transform/quant path alone -- residual against the co-located
    block of the previous frame, forward transform + quant + dequant + inverse + reconstruction
    (x265amd_tq_roundtrip), SLICE_I rounding, no search."""
    import torch.distributed as dist
    frames_np = make_frames(nw, nh, ctu, 16, fps + 1, seed=12345 + 1000 * rank)
    h, w = frames_np.shape[1:]
    frames = torch.from_numpy(frames_np).cuda()
    cur, pred = frames[1:], frames[:-1]
    level = torch.empty((fps, h, w), dtype=torch.int16, device="cuda")
    sums = torch.empty((fps, (h // tu) * (w // tu)), dtype=torch.int32, device="cuda")
    rec = torch.empty((fps, h, w), dtype=torch.uint8, device="cuda")

    def step():
        x265_amd.tq_roundtrip(cur, pred, tu, qp, x265_amd.SLICE_I, out=(level, sums, rec))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ms = e0.elapsed_time(e1) / args.steps
        tq_bytes = 5 * w * h * fps
        out = {
            "metric": "TQ Mpixels/s", "value": round(nw * nh * fps * world * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 8-bit luma (padded {w}x{h}), {tu}x{tu} residual + DCT + quant + dequant + "
                                   f"IDCT + recon, QP{qp}, I-slice rounding, prediction = co-located block of the previous "
                                   f"frame; {fps} frames per step per GPU", "name": args.workload, "tu": tu},
            "roofline": {"kernel": f"tq_kernel<{tu}, 3>", "bound": "hbm", "achieved": round(tq_bytes / (ms * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tq_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": None, "algorithmic_bytes_per_launch": tq_bytes, "avg_launch_ms": round(ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle
            orc = pyoracle.Oracle()
            threads = min(16, os.cpu_count() or 1)
            t0 = time.perf_counter()
            lvl, _ = orc.tq_forward(frames_np[1], frames_np[0], tu, qp, 0, nthreads=threads)
            orec = orc.tq_inverse(lvl, frames_np[0], tu, qp, 0, nthreads=threads)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(nw * nh / dt / 1e6, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port",
                                   "sample": f"one whole frame, forward + inverse, in {dt:.3f} s (oracle port, OpenMP)",
                                   "parity_with_gpu_on_sample": bool((lvl == level[0].cpu().numpy()).all()
                                                                     and (orec == rec[0].cpu().numpy()).all())}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def closed_loop(args, world, rank, backend, nw, nh, rng, nseg, qp):
    """Closed-loop P pictures (x265_amd.ClosedLoopGop): `nseg` independent GOP segments advance one picture per
    step -- search against the previous RECONSTRUCTION, luma 32x32 + chroma 16x16 transform / quant /
    reconstruction, reference rotation.  Pixels counted: luma of the nominal picture, like the main metric."""
    import torch.distributed as dist
    from x265_amd import ClosedLoopGop
    T = 4
    w, h = nw, (nh + 63) // 64 * 64
    segs = []
    for s_ in range(nseg):
        y, u, v = synth.make_sequence(w, h, T, search_range=rng, seed=12345 + 1000 * rank + s_, chroma=True)
        segs.append(np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1))
    yuv = torch.from_numpy(np.stack(segs)).cuda()          # [S][T][W*H*3/2]
    enc = ClosedLoopGop(w, h, nseg, search_range=rng, qp=qp)
    enc.set_reference(yuv[:, 0].contiguous())
    pics = [yuv[:, t].contiguous() for t in range(1, T)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    k = 0
    for _ in range(args.warmup):
        enc.encode_p(pics[k % len(pics)])
        k += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc.encode_p(pics[k % len(pics)])
        k += 1
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "closed-loop ME+TQ Mpixels/s", "value": round(nw * nh * nseg * world * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{nw}x{nh} 4:2:0 (padded {w}x{h}), closed loop: {nseg} GOP segments per GPU advance one P "
                                   f"picture per step; +-{rng} search on the previous reconstruction, luma 32x32 and chroma "
                                   f"16x16 TQ + reconstruction QP{qp}, reference rotation", "name": args.workload,
                       "me_method": enc.method},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--me-method", default="sea", choices=["sea", "brute"],
                    help="sea: exact successive elimination (default); brute: every candidate's SAD. Same output.")
    ap.add_argument("--tu", type=int, default=None, choices=[4, 8, 16, 32], help="transform size (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world, rank, local = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torchrun (one process per GPU); see the module docstring",
                  file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # one process per GPU.  X265AMD_DIST_BACKEND=gloo is for rehearsing the multi-rank path on a box with
    # fewer GPUs than ranks (ranks then share devices); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("X265AMD_DIST_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    x265_amd.init(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    nw, nh, ctu, rng, nrefs, fps, tu, qp = WORKLOADS[args.workload]
    tu = args.tu or tu
    if args.workload == "tq1080p":
        return tq_only(args, world, rank, backend, nw, nh, ctu, fps, tu, qp)
    if args.workload == "gop1080p":
        return closed_loop(args, world, rank, backend, nw, nh, rng, fps, qp)
    frames_np = make_frames(nw, nh, ctu, rng, fps + nrefs, seed=12345 + 1000 * rank)
    h, w = frames_np.shape[1:]
    frames = torch.from_numpy(frames_np).cuda()
    pipe = MeTqPipeline(frames, ctu=ctu, search_range=rng, nrefs=nrefs, tu=tu, qp=qp, method=args.me_method)
    nctu = pipe.nctu

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        pipe.motion_search()
        ev[k][1].record()
        pipe.transform_quant()
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    me_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    tq_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # per-kernel split of the search stage (library-side HIP events on the launch stream), taken right
    # after the timed region on a few more passes of the same batch: the events are read back per pass
    kernels_ms = None
    if pipe.method == "sea" and rank == 0:
        import ctypes
        lib = x265_amd.api._lib()
        lib.x265amd_debug_me_timing.argtypes = [ctypes.c_int]
        lib.x265amd_debug_me_last_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
        lib.x265amd_debug_me_timing(1)
        acc = np.zeros(4)
        reps = min(args.steps, 10)
        buf = (ctypes.c_float * 4)()
        for _ in range(reps):
            pipe.motion_search()
            if lib.x265amd_debug_me_last_ms(buf) != 0:
                acc[:] = np.nan
                break
            acc += np.array(buf[:])
        lib.x265amd_debug_me_timing(0)
        kernels_ms = dict(zip(["box16_pairs", "_", "sea_bounds_kernel", "sea_exact_kernel"],
                              [round(float(x) / reps, 4) for x in acc]))
        kernels_ms.pop("_", None)

    if rank == 0:
        px_step = nw * nh * fps                      # nominal pixels (1920x1080), SURVEY.md 8d
        value = px_step * world * args.steps / elapsed / 1e6
        me_bytes = me_algorithmic_bytes(w, h, nctu, nrefs) * fps
        achieved = me_bytes / (me_ms * 1e-3) / 1e9
        tq_bytes = 5 * w * h * fps                   # src + pred + 2 B level + rec
        sad_ops = 0
        ncx = w // ctu
        for i in range(nctu):  # legal candidates per CTU (clamped at the picture edge)
            cx, cy = (i % ncx) * ctu, (i // ncx) * ctu
            nx = min(rng, cx) + min(rng, w - ctu - cx) + 1
            ny = min(rng, cy) + min(rng, h - ctu - cy) + 1
            sad_ops += nx * ny
        sad_ops *= ctu * ctu * nrefs * fps
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get(args.workload, {}).get("me_bytes_per_launch")
            except Exception:
                traffic = None
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
                                   f"{fps} frames per step per GPU, frames sharded across GPUs, no collective",
                       "name": args.workload, "frames_per_step_per_gpu": fps, "me_method": pipe.method},
            "roofline": {"kernel": "search stage = box16_pairs + sea_bounds_kernel + sea_exact_kernel (dominant)"
                                   if pipe.method == "sea" else "me_fullsearch_qsad",
                         "bound": "hbm", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "algorithmic_bytes_per_launch": me_bytes,
                         "avg_launch_ms": round(me_ms, 4), "kernels_ms": kernels_ms,
                         "note": "avg_launch_ms = the whole search stage per step (sum of its kernels; compare the "
                                 "sum of their averages in profiles/*kernel_stats.csv).  Exhaustive SAD search is "
                                 "integer-VALU bound, not HBM bound (SURVEY.md 8d); see valu.  traffic = PMC "
                                 "FETCH_SIZE (x2, gfx950) + WRITE_SIZE over the stage's kernels "
                                 "(profiles/pmc_traffic.json); it counts fabric requests, Infinity-Cache hits "
                                 "included (window overlap between neighbouring CTUs, box-sum workspace)"},
            "valu": {"pixel_sads_per_launch": sad_ops, "achieved_Tpx_sad_per_s": round(sad_ops / (me_ms * 1e-3) / 1e12, 3),
                     "peak_Tpx_sad_per_s": 152.0,
                     "peak_source": "tools/ubench_valu: v_qsad_pk_u16_u8 issue rate on this chip (profiles/ubench_r1.txt)"},
            "roofline_tq": {"kernel": "tq_roundtrip", "bound": "hbm", "achieved": round(tq_bytes / (tq_ms * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(tq_bytes / (tq_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "algorithmic_bytes_per_launch": tq_bytes, "avg_launch_ms": round(tq_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            gmv = x265_amd.mv_to_numpy(pipe.mv)[0, 0]
            out["cpu_baseline"] = cpu_baseline(frames_np, ctu, rng, nrefs, tu, qp, gmv, pipe.level[0].cpu().numpy(),
                                               pipe.rec[0].cpu().numpy(), nw * nh)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
