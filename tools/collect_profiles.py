#!/usr/bin/env python3
"""Copy the summaries of tools/refresh_profiles.sh (gpurun_out/prof/, scratch) into profiles/ (tracked), named
per round:  python tools/collect_profiles.py r2
  bench lines           -> profiles/<round>_<workload>_bench.json
  rocprofv3 kernel stats-> profiles/<round>_<workload>_kernel_stats.csv
  PMC traffic           -> profiles/pmc_traffic.json (the file bench.py reads) and profiles/<round>_pmc_traffic.json
  SQ counter summaries  -> profiles/<round>_pmc_sq.json, profiles/<round>_pmc_sq_intra1080p.json
  PU search / intra frame timings, TQ wave trace -> profiles/<round>_{pu_bench,intra_bench,tq_trace}.txt"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
    n = 0
    for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
        line = open(f).read().strip().splitlines()[-1]
        json.loads(line)
        name = os.path.basename(f)[len("bench_"):-len(".json")]
        with open(os.path.join(DST, f"{rnd}_{name}_bench.json"), "w") as o:
            o.write(line + "\n")
        n += 1
    for d in sorted(glob.glob(os.path.join(SRC, "trace_*"))):
        if not os.path.isdir(d):
            continue
        name = os.path.basename(d)[len("trace_"):]
        stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if stats:  # earlier refreshes leave their files beside the new ones (named by process id): the newest counts
            shutil.copy(max(stats, key=os.path.getmtime), os.path.join(DST, f"{rnd}_{name}_kernel_stats.csv"))
            n += 1
    p = os.path.join(SRC, "pmc_traffic.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, "pmc_traffic.json"))
        shutil.copy(p, os.path.join(DST, f"{rnd}_pmc_traffic.json"))
        n += 2
    sq = {}
    for p in (os.path.join(ROOT, "gpurun_out", "pmcsq_1080p", "pmc_sq.json"),
              os.path.join(ROOT, "gpurun_out", "pmctq_tq1080p", "pmc_sq.json")):
        if os.path.exists(p):
            sq.update(json.load(open(p)))
    if sq:
        with open(os.path.join(DST, f"{rnd}_pmc_sq.json"), "w") as o:
            json.dump(sq, o, indent=1, sort_keys=True)
        n += 1
    p = os.path.join(ROOT, "gpurun_out", "pmcsq_intra1080p", "pmc_sq.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f"{rnd}_pmc_sq_intra1080p.json"))
        n += 1
    for name in ("pu_bench", "intra_bench", "tq_trace", "encode_i_bench", "sea_stats", "noise_split", "intra_stats", "host_pcie", "gop_steps"):
        p = os.path.join(SRC, name + ".txt")
        if os.path.exists(p):
            with open(p) as i, open(os.path.join(DST, f"{rnd}_{name}.txt"), "w") as o:
                o.writelines(line for line in i if "amdgpu.ids" not in line)
            n += 1
    print(n, "files into", DST)


if __name__ == "__main__":
    main()
