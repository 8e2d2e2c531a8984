#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name, and (with --json OUT) the
HBM-side traffic of the roofline kernel of bench.py in bytes per launch.
    python tools/pmc_summary.py <fetch-dir> <write-dir> [--json profiles/pmc_traffic.json]

Units / corrections (MI355X_MICROARCH.md, HBM section; calibrated in the same passes on tools/pmc_conv.py's 64 MiB
lincomb copy): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide
(16 B / lane) read at 64 B, i.e. half the bytes -- doubled here; WRITE_SIZE matched the known byte count 1:1."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


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
    means = {}
    for p in argv:
        for (kern, ctr), vals in sorted(collect(p).items()):
            k = kern if len(kern) < 110 else kern[:107] + "..."
            means[(kern, ctr)] = sum(vals) / len(vals)
            print("%-14s mean %14.1f  min %14.1f  max %14.1f  n=%4d  %s" % (ctr, sum(vals) / len(vals), min(vals), max(vals), len(vals), k))
    if out_json:
        def pick(sub, ctr):
            for (kern, c), v in means.items():
                if c == ctr and sub in kern:
                    return v
            return None
        f, w = pick("conv3x3_tile_kernel", "FETCH_SIZE"), pick("conv3x3_tile_kernel", "WRITE_SIZE")
        cf, cw = pick("lincomb_kernel", "FETCH_SIZE"), pick("lincomb_kernel", "WRITE_SIZE")
        res = {}
        if f is not None and w is not None:
            res["conv3x3_tile_kernel@[4,32,32,64->64]_bf16"] = {
                "fetch_bytes": int(f * 1024 * 2), "write_bytes": int(w * 1024),
                "raw_FETCH_SIZE_KiB": f, "raw_WRITE_SIZE_KiB": w, "fetch_correction": 2.0,
                "calibration": {"kernel": "lincomb 2 x 64 MiB in, 64 MiB out", "FETCH_SIZE_KiB": cf, "WRITE_SIZE_KiB": cw,
                                "expected_read_KiB": 131072, "expected_write_KiB": 65536}}
        os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
        with open(out_json, "w") as fh:
            json.dump(res, fh, indent=1)
        print("wrote", out_json)


if __name__ == "__main__":
    main(sys.argv[1:])
