#!/usr/bin/env python3
"""Kernel durations and the idle gaps between them from a rocprofv3 --kernel-trace CSV (last N dispatches):
   python tools/trace_gaps.py <dir-or-csv> [N]"""
import csv
import sys
from pathlib import Path


def main():
    src = Path(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    f = src if src.is_file() else next(iter(sorted(src.rglob("*kernel_trace.csv"))))
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev = None
    busy = idle = 0.0
    for r in rows[-n:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1000 if prev else 0.0
        busy += (e - s) / 1000
        idle += max(gap, 0.0)
        name = r["Kernel_Name"][:44]
        grid = r.get("Grid_Size", "")
        print(f"{name:44s} dur {(e - s) / 1000:8.2f} us  gap {gap:8.2f} us  grid {grid}")
        prev = e
    print(f"busy {busy:.1f} us, idle between kernels {idle:.1f} us")


if __name__ == "__main__":
    main()
