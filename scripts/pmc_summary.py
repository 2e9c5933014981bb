"""Summarise rocprofv3 --pmc CSV passes (scripts/profile_pmc.sh) per kernel: mean counter value
per dispatch. usage: python scripts/pmc_summary.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{root}/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/stats/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# per-dispatch means from {root} (rocprofv3 --pmc, one counter group per pass; durations from the --stats pass)")
for k in sorted(agg):
    if k.startswith("__amd"):
        continue
    print(f"\n## {k}   launches={len(dur.get(k, []))}  avg_duration_us={sum(dur.get(k, [0])) / max(1, len(dur.get(k, []))):.1f}")
    for c, v in sorted(agg[k].items()):
        print(f"  {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    if "FETCH_SIZE" in c:
        # FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md)
        print(f"  -> HBM-side read  = {c['FETCH_SIZE'] * 1024 / 1e6:.1f} MB as reported, {2 * c['FETCH_SIZE'] * 1024 / 1e6:.1f} MB with the gfx950 x2 correction")
    if "WRITE_SIZE" in c:
        print(f"  -> HBM-side write = {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB (uncalibrated)")
    if "TCC_HIT_sum" in c:
        print(f"  -> L2 hit rate    = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"]:
        # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"  -> MFMA pipe busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} of SIMD-cycles "
              f"(dispatch = {cyc:.4g} shader cycles, effective clock {cyc / (sum(dur[k]) / len(dur[k])) / 1e3:.2f} GHz)")
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
        print(f"  -> VALU instr (incl. MFMA) {c['SQ_INSTS_VALU']:.3g}, MFMA {c['SQ_INSTS_MFMA']:.3g}")
