"""Dev tool: prints the kernels of the LAST all-pairs pass in a rocprofv3 --kernel-trace csv with their start offsets, durations and
the gaps between them. usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the probe starts the pass
idx = max(i for i, r in enumerate(rows) if "k_prefilter_probe" in r["Kernel_Name"])
first = max(0, idx - 2)
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:idx + 6]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:10.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
