"""profiles/hbm_traffic.json from a round's PMC summaries (scripts/profile_round.sh + scripts/profile_rgb.sh): bytes per launch of the
dominant kernels = FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024, each from its own pass. The
file is keyed to the kernel sources it was measured on (`_kernel_sources_sha256`): bench.py reports `traffic_stale` when a kernel
file has changed since.
  python scripts/make_hbm_traffic.py r06        (reads gpurun_out/r06_pmc_*.txt and gpurun_out/pmc_r06_rgb summary)"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hydrus-video-deduplicator_amd", "csrc")
KERNEL_FILES = ("k_hamming_mfma.hip", "k_hamming.hip", "k_pdq.hip", "k_vmatch.hip")


def kernel_sources_sha256():
    return {f: hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16] for f in KERNEL_FILES}


def sections(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"## (.+?)\s+launches=(\d+)\s+avg_duration_us=([\d.]+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {"launches": int(m.group(2)), "us": float(m.group(3))})
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def traffic(sec):
    return int(round((sec.get("FETCH_SIZE", 0.0) * 2 + sec.get("WRITE_SIZE", 0.0)) * 1024, -5))


def main(tag):
    go = os.path.join(ROOT, "gpurun_out")
    new = {}
    s = sections(os.path.join(go, f"{tag}_pmc_summary.txt"))
    ap = max((v for k, v in s.items() if k.startswith("k_allpairs_mfma")), key=lambda v: v["us"])
    new["allpairs_n1000000_v13_w1"] = new["allpairs_n1000000_v9_w1"] = traffic(ap)
    for n in (10000, 400000):
        s = sections(os.path.join(go, f"{tag}_pmc_k1_{n}.txt"))
        new[f"pdq_hash64_n{n}"] = traffic(next(v for k, v in s.items() if k.startswith("k_pdq_hash64")))
    s = sections(os.path.join(go, f"{tag}_pmc_cfg5.txt"))
    ap5 = max((v for k, v in s.items() if k.startswith("k_allpairs_mfma")), key=lambda v: v["us"])
    new["config5_search_v50000x64_w1"] = traffic(ap5)
    rgb = os.path.join(go, f"{tag}_pmc_down512w.txt")
    if os.path.exists(rgb):
        s = sections(rgb)
        new["down512w_rgb_n6144"] = traffic(next(v for k, v in s.items() if k.startswith("k_down512w<3>")))
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    old = json.load(open(path))
    prev_round = {k: v for k, v in old.items() if not k.startswith("_")}
    out = {"_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/profile_round.sh {tag}, scripts/profile_rgb.sh, "
                      "scripts/pmc_summary.py), bytes per launch; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md "
                      f"(KiB units). Summaries: profiles/{tag}_pmc_summary.txt, {tag}_pmc_k1_*.txt, {tag}_pmc_cfg5.txt, {tag}_pmc_down512w.txt",
           "_round": tag, "_kernel_sources_sha256": kernel_sources_sha256()}
    out.update(new)
    for k, v in old.items():
        if k.startswith("_round") and k != "_round":
            out[k] = v
    out[f"_round{int(tag[1:]) - 1}"] = prev_round
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_round") or k == "_round"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
