"""Timeline of one step from a rocprofv3 --kernel-trace CSV: kernels in start order with the idle gap
before each.   python tools/trace_gaps.py <kernel_trace.csv> [first_kernel_substring] [n_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "prep_wave_kernel<unsigned char"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[len(starts) // 2]  # a step in the middle of the run
prev_end = int(rows[i0 - 1]["End_Timestamp"]) if i0 else int(rows[i0]["Start_Timestamp"])
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
