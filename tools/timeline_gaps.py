#!/usr/bin/env python3
"""Gaps between consecutive kernels of the timed sweeps, from a rocprofv3 --kernel-trace CSV.
    python tools/timeline_gaps.py <dir with *_kernel_trace.csv> [last N sweeps]"""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**/*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bca_sweep_csr_kernel" in r["Kernel_Name"]][-last:]
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    parts = []
    for p, q in zip(seg[:-1], seg[1:]):
        parts.append(f"{p['Kernel_Name'].split('(')[0].split('::')[-1][:22]} {(int(p['End_Timestamp']) - int(p['Start_Timestamp'])) / 1e3:.1f}us"
                     f" |gap {(int(q['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3:.1f}us|")
    print(f"step {(int(seg[-1]['Start_Timestamp']) - t0) / 1e3:.1f}us: " + " ".join(parts))
