#!/usr/bin/env python3
"""Aggregate the SQ counter passes of tools/run_pmc_sq.sh into profiles/r1_pmc_sq.json: per kernel, the
raw counters averaged per dispatch and the usual derived figures (gfx94x formulas; SQ cycle counters are
in quad-cycles, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    python tools/pmc_sq.py <workload> <dir with a/ b/ c/> [out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SIMDS, CUS, XCDS = 1024, 256, 8


def load(d):
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("x265amd::", "").replace("void ", "")
                tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
                n[name][row["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}


def main():
    workload, d = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "r1_pmc_sq.json")
    merged = defaultdict(dict)
    for sub in ("a", "b", "c", "d"):
        for k, v in load(os.path.join(d, sub)).items():
            merged[k].update(v)
    res = {}
    for k, c in sorted(merged.items()):
        if not k.startswith(("sea_", "box16", "tq_kernel", "tq_mfma", "me_fullsearch", "me_list_qsad", "me_tree", "sad_blocks", "intra_")):
            continue
        # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (sea_exact_kernel: 9.9 M for a 0.49 ms kernel = 8 x 1.24 M
        # cycles); the SQ counters are chip totals (SQ_WAVES = every wave of the grid)
        g = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        der = {}
        if g and "SQ_ACTIVE_INST_VALU" in c:
            der["VALUBusy_pct"] = 100.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / g
        if c.get("SQ_ACTIVE_INST_VALU"):
            der["VALUUtilization_pct"] = 100.0 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
        if c.get("SQ_WAVES"):
            der["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            if "SQ_INSTS_LDS" in c:
                der["lds_insts_per_wave"] = c["SQ_INSTS_LDS"] / c["SQ_WAVES"]
            if "SQ_INSTS_SMEM" in c:
                der["smem_insts_per_wave"] = c["SQ_INSTS_SMEM"] / c["SQ_WAVES"]
        if g and "SQ_WAVE_CYCLES" in c:
            der["mean_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (g * SIMDS)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            der["LDSBankConflict_pct_of_lds_cycles"] = 100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if g and "SQ_LDS_IDX_ACTIVE" in c:
            der["lds_busy_pct"] = 100.0 * c["SQ_LDS_IDX_ACTIVE"] / (g * CUS)
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
            w = c["SQ_WAVE_CYCLES"]
            der["wave_cycles_split_pct"] = {"wait_any": 100 * c["SQ_WAIT_ANY"] / w,
                                            "wait_inst_any": 100 * c["SQ_WAIT_INST_ANY"] / w,
                                            "active_inst_any": 100 * c["SQ_ACTIVE_INST_ANY"] / w}
        res[k] = {"counters_per_dispatch": c, "derived": der}
    data = {}
    if os.path.exists(out):
        with open(out) as f:
            data = json.load(f)
    data[workload] = {"kernels": res,
                      "note": "rocprofv3 --pmc, three separate passes (tools/run_pmc_sq.sh); averages per dispatch; "
                              "GRBM_GUI_ACTIVE divided by 8 (it is summed over the XCDs); derived with the gfx94x formulas (VALUBusy = ACTIVE_INST_VALU*4/SIMDs/GUI_ACTIVE, "
                              "VALUUtilization = THREAD_CYCLES_VALU/(ACTIVE_INST_VALU*64)); SQ cycle counters in quad-cycles"}
    with open(out, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    for k, v in res.items():
        print(k, json.dumps(v["derived"]))


if __name__ == "__main__":
    main()
