#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: median / min / max per counter and dispatch of one kernel.

usage: pmc_summary.py <rocprof output dir> [kernel-name substring] [--json profiles/pmc_traffic.json]
Walks the directory for *counter_collection.csv (one file per pass)."""
import csv
import json
import os
import statistics
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = args[0]
    needle = args[1] if len(args) > 1 else "gp_fit_fused"
    vals = {}
    for dp, _, files in os.walk(root):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            with open(os.path.join(dp, f)) as fh:
                for row in csv.DictReader(fh):
                    if needle not in row.get("Kernel_Name", ""):
                        continue
                    vals.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                    vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out = {}
    for name in sorted(vals):
        v = list(vals[name].values())
        out[name] = statistics.median(v)
        print(f"{name:32s} n={len(v):3d} median={out[name]:16.1f} min={min(v):16.1f} max={max(v):16.1f}")
    if "--json" in sys.argv:
        path = sys.argv[sys.argv.index("--json") + 1]
        if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
            # gfx950: FETCH_SIZE reports half the bytes read (MI355X_MICROARCH.md; confirmed for 8- and 16-byte per-lane
            # streams by tools/fetch_calib.hip, profiles/r02_fetch_write_size_calibration.txt); WRITE_SIZE is exact
            rec = dict(hbm_bytes_per_launch=(2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, fetch_size_kb_raw=out["FETCH_SIZE"],
                       fetch_size_kb_corrected=2.0 * out["FETCH_SIZE"], write_size_kb=out["WRITE_SIZE"],
                       note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), median per dispatch of "
                            f"{needle}; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)")
            with open(path, "w") as fh:
                json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
