#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of the bench workload into profiles/<tag>_pmc.json, keyed by the launch profiler's
kernel names (tools/knames.py), per launch:
    fetch_bytes  = FETCH_SIZE [KiB] * 1024 * 2      (gfx950: FETCH_SIZE tallies the 128-B requests of wide reads at 64 B,
                                                     MI355X_MICROARCH.md section HBM; calibrated in the same pass on a
                                                     64 MiB-in streaming kernel when the run contains tg lincomb launches)
    write_bytes  = WRITE_SIZE [KiB] * 1024
    hbm_bytes_per_launch = fetch_bytes + write_bytes
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
        SQ_VALU_MFMA_BUSY_CYCLES is the sum over all MFMA instructions of their pipe cycles (calibrated: the chain kernel
        <4,16> at [4,32,32,64->64] issues 256 WG x 4 waves x 18 v_mfma_f32_16x16x32_bf16 = 18432 instructions x 16 cycles
        = 294912, exactly the counter's value); GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 = SIMDs of the chip.
        For kernels of a few microseconds GUI_ACTIVE also holds the profiler's serialisation gaps (the fraction is a lower
        bound there).
Counters come from SEPARATE passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains).

    python tools/pmc_summary.py --json profiles/r02_pmc.json <dir-or-csv> [...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from knames import tg_name  # noqa: E402


def collect(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


def main(argv):
    out_json = None
    if "--json" in argv:
        i = argv.index("--json")
        out_json = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    per = defaultdict(lambda: defaultdict(list))
    for p in argv:
        for (kern, ctr), vals in collect(p).items():
            name = tg_name(kern) or kern.split("(")[0][:80]
            per[name][ctr].extend(vals)
    res = {}
    for name, ctrs in sorted(per.items()):
        e = {"launches_counted": max(len(v) for v in ctrs.values())}
        mean = {c: sum(v) / len(v) for c, v in ctrs.items()}
        if "FETCH_SIZE" in mean:
            e["raw_FETCH_SIZE_KiB"] = round(mean["FETCH_SIZE"], 2)
            e["fetch_bytes"] = int(mean["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in mean:
            e["raw_WRITE_SIZE_KiB"] = round(mean["WRITE_SIZE"], 2)
            e["write_bytes"] = int(mean["WRITE_SIZE"] * 1024)
        if "fetch_bytes" in e and "write_bytes" in e:
            e["hbm_bytes_per_launch"] = e["fetch_bytes"] + e["write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            e["raw_SQ_VALU_MFMA_BUSY_CYCLES"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"], 1)
            e["raw_GRBM_GUI_ACTIVE"] = round(mean["GRBM_GUI_ACTIVE"], 1)
            e["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 5)
        for c in ("SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            if c in mean:
                e["raw_" + c] = round(mean[c], 1)
        res[name] = e
        print("%-44s %s" % (name, {k: v for k, v in e.items() if not k.startswith("raw_")}))
    if out_json:
        os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
        with open(out_json, "w") as fh:
            json.dump(res, fh, indent=1, sort_keys=True)
        print("wrote", out_json)


if __name__ == "__main__":
    main(sys.argv[1:])
