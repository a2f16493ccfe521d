#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <workload> <launches> <fetch_dir> <write_dir> [out.json]

The counters are averaged per DISPATCH of each kernel; every kernel of the path is dispatched
once per step, so that is bytes per launch (`launches` = warmup + steps is recorded for reference;
bench.py's per-kernel timing passes add dispatches, which is why the average is per dispatch).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB-like units of 1 KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads, so the fetch figure is given raw and doubled (`fetch_bytes_x2`), and the doubled
figure is what `me_bytes_per_launch` uses (an upper estimate for narrow reads).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    tot = defaultdict(float)
    n = defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("x265amd::", "").replace("void ", "")
                tot[name] += float(row["Counter_Value"])
                n[name] += 1
    return tot, n


def main():
    workload, launches, fdir, wdir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    fetch, nf = load(fdir, "FETCH_SIZE")
    write, nw = load(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        kernels[k] = {
            "dispatches": nf.get(k, 0) or nw.get(k, 0),
            "fetch_bytes_raw": fetch.get(k, 0.0) * 1024 / max(nf.get(k, 0), 1),
            "fetch_bytes_x2": fetch.get(k, 0.0) * 2048 / max(nf.get(k, 0), 1),
            "write_bytes": write.get(k, 0.0) * 1024 / max(nw.get(k, 0), 1),
        }
    me = [k for k in kernels if k.startswith(("sea_", "box16", "me_fullsearch", "me_list_qsad"))]
    tq = [k for k in kernels if k.startswith(("tq_kernel", "tq_mfma"))]
    entry = {
        "launches": launches,
        "kernels": kernels,
        "me_bytes_per_launch": sum(kernels[k]["fetch_bytes_x2"] + kernels[k]["write_bytes"] for k in me),
        "me_bytes_per_launch_raw_fetch": sum(kernels[k]["fetch_bytes_raw"] + kernels[k]["write_bytes"] for k in me),
        "tq_bytes_per_launch": sum(kernels[k]["fetch_bytes_x2"] + kernels[k]["write_bytes"] for k in tq),
        "tq_bytes_per_launch_raw_fetch": sum(kernels[k]["fetch_bytes_raw"] + kernels[k]["write_bytes"] for k in tq),
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; KB -> bytes; gfx950 "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (raw figure kept beside it)",
    }
    data = {}
    if os.path.exists(out):
        with open(out) as f:
            data = json.load(f)
    data[workload] = entry
    with open(out, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in entry.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
