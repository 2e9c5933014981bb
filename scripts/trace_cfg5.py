"""Dev tool: timeline of the last config-5 video search in a rocprofv3 --kernel-trace csv (from the last k_pack_fp4 to the end)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = max(range(len(rows)), key=lambda i: (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]), i))
first = max(0, big - 8)
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:big + 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:10.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:80]}")
    prev_end = e
