#!/usr/bin/env python3
"""Post-processes rocprofv3 output into the summaries kept under profiles/.

    python tools/profile_summary.py stats  <dir of --kernel-trace --stats run>   <out.csv>
    python tools/profile_summary.py hbm    <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> <out.json> [train steps of the run]
    python tools/profile_summary.py pmc    <dir of a --pmc run> <out.csv>

hbm: FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE tallies the 128-B requests of wide
streaming reads at 64 B, so it is doubled (MI355X_MICROARCH.md, section HBM) - the uncorrected value is kept beside it.
Per kernel: launches and the per-launch average."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0] if "<" in name or "(" in name else name


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    mode = sys.argv[1]
    if mode == "stats":
        rows = []
        for f in find(sys.argv[2], "*kernel_stats.csv"):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        with open(sys.argv[3], "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
            for r in rows:
                if float(r["Percentage"]) < 0.05:
                    continue
                w.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.1f}',
                            r["Percentage"], f'{float(r["MinNs"]) / 1e3:.1f}', f'{float(r["MaxNs"]) / 1e3:.1f}'])
    elif mode == "hbm":
        fe, wr = counters(sys.argv[2]), counters(sys.argv[3])
        out = {}
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, {}).get("FETCH_SIZE", [0, 0])[1])):
            n, fs = fe.get(k, {}).get("FETCH_SIZE", [0, 0.0])
            n2, ws = wr.get(k, {}).get("WRITE_SIZE", [0, 0.0])
            if max(n, n2) == 0 or (fs + ws) * 1024 < 50e6:
                continue
            out[k] = {"launches": max(n, n2), "fetch_bytes_per_launch_reported": fs * 1024 / max(n, 1),
                      "fetch_bytes_per_launch_corrected": 2 * fs * 1024 / max(n, 1), "write_bytes_per_launch": ws * 1024 / max(n2, 1)}
        if len(sys.argv) > 5:      # optimiser steps the profiled command ran (warm-up + timed + bench.py's three profiled passes)
            out["_run"] = {"train_steps": int(sys.argv[5])}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
    elif mode == "pmc":
        acc = counters(sys.argv[2])
        names = sorted({c for v in acc.values() for c in v})
        with open(sys.argv[3], "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "dispatches"] + names)
            for k, v in sorted(acc.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
                n = max(x[0] for x in v.values())
                if n == 0:
                    continue
                w.writerow([k, n] + [f"{v[c][1] / max(v[c][0], 1):.1f}" if c in v else "" for c in names])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
