"""Code-shape guard (CPU test): the kernels' speed depends on budgets that nothing else in the suite
reads -- VGPRs per lane (resident waves per SIMD), LDS per workgroup (resident workgroups per CU), spills, and the
presence of the instructions the design is built on. The all-pairs kernel's own measurements: 1 / 2 / 3 resident waves =
40.6 / 23.0 / 18.2 ms, and the default forms sit exactly on the 3-wave limit (168 VGPRs; 3 x 53 344 B of LDS of 163 840).
A compiler bump or a two-register change would cost 25 % silently; only a bench run would show it.

hipcc cross-compiles the two kernel files for gfx950 with --cuda-device-only -S (no GPU needed, ~10 s) and this module
reads every kernel's `.amdhsa` metadata and ISA text. Budget table: DESIGN.md section 4.5 ("Code shape").
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hydrus-video-deduplicator_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

VGPRS_PER_SIMD_LANE = 512   # gfx950: unified VGPR file, 512 registers per lane per SIMD
VGPR_GRANULE = 8
LDS_PER_CU = 160 * 1024


def waves_per_simd(vgprs: int) -> int:
    alloc = (vgprs + VGPR_GRANULE - 1) // VGPR_GRANULE * VGPR_GRANULE
    return min(8, VGPRS_PER_SIMD_LANE // alloc)


def _makefile_flags():
    """The very flags the product build uses (csrc/Makefile: CXXFLAGS), so that this test sees the product's code."""
    txt = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^CXXFLAGS\s*:=\s*(.+)$", txt, re.M)
    flags = m.group(1).replace("$(ARCH)", re.search(r"^ARCH\s*:=\s*(\S+)", txt, re.M).group(1)).split()
    assert "--offload-arch=gfx950" in flags
    return flags


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "").replace("void ", "")).strip() for d in out[: len(names)]]


def _compile(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    subprocess.run([HIPCC] + _makefile_flags() + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                   check=True, capture_output=True, text=True)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    kernels = {}
    blocks = meta.split("  - .agpr_count:")[1:]
    mangled = [re.search(r"\.name:\s+(\S+)", b).group(1) for b in blocks]
    for b, mg, name in zip(blocks, mangled, _demangle(mangled)):
        def num(key, b=b):
            return int(re.search(rf"\.{key}:\s+(\d+)", b).group(1))

        start = text.index(f"\n{mg}:")
        end = text.index(".Lfunc_end", start)
        kernels[name] = {"vgpr": num("vgpr_count"), "agpr": int(re.match(r"\s*(\d+)", b).group(1)),
                         "vgpr_spill": num("vgpr_spill_count"), "sgpr_spill": num("sgpr_spill_count"),
                         "lds": num("group_segment_fixed_size"), "scratch": num("private_segment_fixed_size"),
                         "wg": num("max_flat_workgroup_size"), "isa": text[start:end]}
    return kernels


@pytest.fixture(scope="module")
def shapes(tmp_path_factory):
    if not (os.path.exists(HIPCC) and shutil.which("c++filt")):
        pytest.fail("hipcc / c++filt missing: the code-shape guard cannot run (it must, on the build container)")
    tmp = str(tmp_path_factory.mktemp("code_shape"))
    return {"mfma": _compile("k_hamming_mfma.hip", tmp), "pdq": _compile("k_pdq.hip", tmp), "img": _compile("k_fp4_image.hip", tmp)}


# form -> template arguments <TILES, NBR, S1, RECT, QUEUE> (k_hamming_mfma.hip: launch table), what the form must keep:
# waves = resident waves per SIMD, spill = VGPR spills allowed (0 everywhere), lds = bytes.
# Round 6: the table holds only the forms that have a job (the eleven measured-and-lost ones are in HISTORY.md).
FORMS = {
    8: ("8, 4, 4, {r}, false", dict(waves=2, spill=0, lds=40992)),   # full 256-bit compare, no first stage: the reference form
    9: ("8, 2, 2, {r}, false", dict(waves=3, spill=0, lds=40992)),   # fetch form: the probe's pick for uniform DBs (headline)
    12: ("4, 4, 2, {r}, false", dict(waves=3, spill=0, lds=40992)),  # register cascade: the probe's pick for dense DBs
    18: ("8, 2, 2, {r}, true", dict(waves=3, spill=0, lds=53344)),   # panel-mark queue: the probe's pick for frame hashes
}
DEFAULT_FORMS = (9, 12, 18)  # what the auto variant (13) can run


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("rect", ["false", "true"])
def test_allpairs_form_keeps_its_occupancy(shapes, form, rect):
    targs, want = FORMS[form]
    k = shapes["mfma"][f"k_allpairs_mfma<{targs.format(r=rect)}>"]
    assert k["agpr"] == 0
    assert waves_per_simd(k["vgpr"]) >= want["waves"], f"form {form}: {k['vgpr']} VGPRs = {waves_per_simd(k['vgpr'])} waves per SIMD"
    assert k["vgpr_spill"] <= want["spill"], f"form {form}: {k['vgpr_spill']} VGPRs spilled"
    assert k["scratch"] <= (48 if want["spill"] else 16)  # 16 B: the frame of the out-of-line drain (no spill traffic in the loop)
    assert k["lds"] <= want["lds"], f"form {form}: {k['lds']} B of LDS"
    # a 256-lane workgroup puts one wave on every SIMD: resident workgroups per CU = resident waves per SIMD
    waves_per_wg_per_simd = k["wg"] // 64 // 4
    assert k["lds"] * (want["waves"] // waves_per_wg_per_simd) <= LDS_PER_CU, f"form {form}: LDS caps the occupancy below {want['waves']}"
    # the design's instructions: FP4 MFMA with block scaling operands, panels fetched straight into LDS
    assert "v_mfma_scale_f32_32x32x64_f8f6f4" in k["isa"] or "v_mfma_f32_32x32x64_f8f6f4" in k["isa"]
    assert "cbsz:4" in k["isa"] and "blgp:4" in k["isa"]  # both operands FP4
    assert "global_load_lds_dwordx4" in k["isa"]
    assert "scratch_load" not in k["isa"] or want["spill"] > 0


def test_default_forms_have_no_spill_traffic_in_the_loop_and_three_waves(shapes):
    """Forms 9, 12 and 18 -- what the probe can pick -- spill nothing and keep three resident waves per SIMD (until round 5
    form 12 parked two loop-invariant LDS addresses in scratch; without the operand-scale parameter of the removed fp4_code
    experiment it fits its 168 registers)."""
    for form in DEFAULT_FORMS:
        for rect in ("false", "true"):
            k = shapes["mfma"][f"k_allpairs_mfma<{FORMS[form][0].format(r=rect)}>"]
            assert k["vgpr"] <= 168, (form, rect, k["vgpr"])
            assert k["vgpr_spill"] == 0 and "scratch_" not in k["isa"], (form, rect, k["vgpr_spill"])


def test_every_allpairs_instantiation_is_in_the_table(shapes):
    built = {n for n in shapes["mfma"] if n.startswith("k_allpairs_mfma<")}
    table = {f"k_allpairs_mfma<{t.format(r=r)}>" for t, _ in FORMS.values() for r in ("false", "true")}
    assert built == table, built ^ table


def test_probe_and_image_kernels(shapes):
    m = shapes["mfma"]
    # (round 5: a third survivor count -- 137 VGPRs = 3 waves per SIMD; capped at 128 it spills 17, and the probe is 0.3 % of a pass)
    assert waves_per_simd(m["k_prefilter_probe"]["vgpr"]) >= 3 and m["k_prefilter_probe"]["vgpr_spill"] == 0
    for name in ("k_expand_fp4", "k_pack_fp4"):  # (csrc/k_fp4_image.hip since round 6)
        k = shapes["img"][name]
        assert waves_per_simd(k["vgpr"]) == 8 and k["lds"] == 0 and k["vgpr_spill"] == 0


# k_pdq_hash64<KIND, DLDS, LUT, PREF>: KIND 0 = u8 gray in, 1 = float luma in (after the down-sampler); DLDS 2 = DCT matrix as
# literals (>= 8192 frames), 0 = SGPRs; launch_pdq_hash64 picks <0,2,1> / <0,0,1> for gray and <1,2,0> / <1,0,0> for luma.
PDQ = {
    "k_pdq_hash64<0, 2, 1, false>": dict(waves=5),  # configs[1] (10 k frames) and everything larger
    "k_pdq_hash64<0, 0, 1, false>": dict(waves=5),
    "k_pdq_hash64<0, 0, 0, false>": dict(waves=5),
    "k_pdq_hash64<0, 0, 2, false>": dict(waves=5),
    "k_pdq_hash64<0, 1, 1, false>": dict(waves=4),
    "k_pdq_hash64<1, 2, 0, false>": dict(waves=5),  # behind k_down512w (the reference's 512x512 frames)
    "k_pdq_hash64<1, 1, 0, false>": dict(waves=4),
    "k_pdq_hash64<1, 0, 0, false>": dict(waves=4),
    "k_pdq_hash64<0, 2, 1, true>": dict(waves=4),   # prefetch experiment (off)
    "k_pdq_hash64<0, 0, 1, true>": dict(waves=4),
    "k_pdq_hash64_fma<0>": dict(waves=3),
    "k_pdq_hash64_fma<1>": dict(waves=3),
    "k_down512w<3>": dict(waves=3),                 # one wave per RGB frame: 12 waves per CU
    "k_down512w<1>": dict(waves=4),
    "k_down512<3, 32>": dict(waves=4),              # 512 lanes x 2 workgroups per CU
    "k_down512<1, 32>": dict(waves=4),
    "k_down512<3, 64>": dict(waves=2),              # round 5: 64-column strips, ONE workgroup per CU (149.5 KB of LDS): batches <= 256 frames
    "k_down512<1, 64>": dict(waves=2),
}


@pytest.mark.parametrize("name", sorted(PDQ))
def test_pdq_kernel_keeps_its_occupancy(shapes, name):
    k = shapes["pdq"][name]
    assert waves_per_simd(k["vgpr"]) >= PDQ[name]["waves"], f"{name}: {k['vgpr']} VGPRs"
    assert k["vgpr_spill"] == 0 and k["scratch"] == 0
    wgs = PDQ[name]["waves"] * 4 // (k["wg"] // 64)  # workgroups per CU at that occupancy
    assert k["lds"] * max(1, wgs) <= LDS_PER_CU or name.startswith("k_pdq_hash64"), (name, k["lds"], wgs)


def test_hash_kernel_lds_allows_seven_workgroups_per_cu(shapes):
    """launch_pdq_hash64 sizes its grid for 7 resident workgroups per CU (the LDS limit the launch code quotes)."""
    for name, k in shapes["pdq"].items():
        if name.startswith("k_pdq_hash64<"):
            assert 7 * k["lds"] <= LDS_PER_CU, (name, k["lds"])


def test_strict_hash_kernel_has_no_fused_multiply_add(shapes):
    """Bit-exactness of the default DCT mode rests on separately rounded multiplies and adds (-ffp-contract=off and the
    explicit __fmul_rn / __fadd_rn): the strict kernels' ISA must hold no FMA/MAC and no MFMA; the opt-in fma kernels must
    run on v_mfma_f32_16x16x4_f32. Three uses of an fma are not contractions of reference arithmetic and are recognised:
    hipcc's expansion of a 64-bit integer division (v_fmamk with +-2^32), its expansion of an IEEE float division
    (v_div_scale .. v_div_fmas .. v_div_fixup: 3 v_fma + 2 v_fmac per division, correctly rounded as a whole) and the
    quality term's exact remainder in the float-luma kernels (grad_term)."""
    fma = re.compile(r"\bv_(?:fma|fmac|mad|mac|pk_fma|dot2c?|mfma)\w*f(?:32|16)\w*")
    for name, k in shapes["pdq"].items():
        lines = [ln.strip() for ln in k["isa"].splitlines()]
        if name.startswith(("k_pdq_hash64<", "k_down512", "k_box_scan_T")) or name == "k_luma64_rgb":
            bad = [ln for ln in lines if fma.search(ln)
                   and not re.search(r"v_fmamk_f32 .*0x[4c]f800000", ln)
                   and not (name.startswith("k_pdq_hash64<1,") and re.fullmatch(r"v_fma_f32 v\d+, v\d+, s\d+, \|v\d+\|", ln))]
            divisions = sum(ln.startswith("v_div_fmas_f32") for ln in lines)
            if name.startswith("k_pdq_hash64<"):
                assert divisions == 0, name  # nothing in the 64x64 kernels divides
            assert len(bad) == 5 * divisions, (name, divisions, sorted(set(bad))[:8])
            assert all(re.match(r"v_fma_f32 v\d+, -v\d+, v\d+, (?:v\d+|1\.0)$|v_fmac_f32_e32 ", ln) for ln in bad), (name, bad[:8])
        if name.startswith("k_pdq_hash64_fma"):
            assert any("v_mfma_f32_16x16x4_f32" in ln or "v_mfma_f32_16x16x4f32" in ln for ln in lines), name


def test_ablation_builds_need_a_second_define(tmp_path):
    """Wrong-result ablation switches (HVD_ABL_*; the all-pairs kernel's were removed with the pruned forms in round 6) must
    not compile out of the product source with one -D: without -DHVD_DEV_ABLATION the preprocessor
    stops with #error."""
    for src, define in (("k_pdq.hip", "-DHVD_ABL_NOFETCH"),):
        r = subprocess.run([HIPCC] + _makefile_flags() + [define, "--cuda-host-only", "-E", os.path.join(CSRC, src), "-o",
                                                          str(tmp_path / "x.ii")], capture_output=True, text=True)
        assert r.returncode != 0 and "developer ablation builds" in r.stderr, (src, define, r.stderr[-300:])
