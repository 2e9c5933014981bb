"""Round 6 GPU tests: the attribution tooling VERDICT r5 item 2 asked for (in-kernel clock telemetry, sysfs read-outs, policy
labels in the runtime block) and the group re-arm entry of ABI 6 on a single context. The full-size oracle checks of the round
live in tests/test_gpu_round5.py (configs[3] row bands, configs[4] every frame and every record)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _get(gpu, key):
    v = C.c_int(0)
    gpu.check(gpu.load().hvd_debug_get(key, C.byref(v)))
    return v.value


@pytest.mark.parametrize("variant", [9, 18, 12, 8])
def test_clock_telemetry_reports_a_plausible_shader_clock_and_never_changes_results(gpu, hvd, oracle, variant):
    """One workgroup in eight of k_allpairs_mfma brackets its lifetime with s_memtime / s_memrealtime (start values in LDS, end
    in flush_pairs_wg); hvd_debug_get "mfma_pass_khz" = cycles / ticks x the runtime's wall-clock rate since the last
    "mfma_clock_reset". Every form samples; a reset clears; the popcount kernel contributes nothing; the pair list is the oracle's."""
    lib = gpu.load()
    n = 120_000
    db, _ = hvd.synth.hash_db(n, seed=61, plant_fraction=0.01)
    want = oracle.allpairs(db, 31, num_threads=8)
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        gpu.check(lib.hvd_debug_set(b"mfma_clock_reset", 1))
        assert _get(gpu, b"mfma_clock_samples") == 0 and _get(gpu, b"mfma_pass_khz") == 0
        got = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=variant)
        assert np.array_equal(got, want)
        samples, khz = _get(gpu, b"mfma_clock_samples"), _get(gpu, b"mfma_pass_khz")
        rows, chunk = hvd.multigpu.tile_geometry(n, variant)
        tiles = int(hvd.multigpu.rank_work_shares(n, 1, variant, tiles=True)[0])
        assert 0 < samples <= tiles and samples >= tiles // 16, (samples, tiles)  # one in eight of the workgroups with work
        assert 1_000_000 <= khz <= 2_500_000, khz  # MI355X: 2.4 GHz peak engine clock; power-limited passes run at 1.9 - 2.2
        # accumulates over passes, clears on reset
        hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=variant)
        assert _get(gpu, b"mfma_clock_samples") == 2 * samples
        gpu.check(lib.hvd_debug_set(b"mfma_clock_reset", 1))
        hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=1)  # integer popcount kernel: no telemetry
        assert _get(gpu, b"mfma_clock_samples") == 0
    finally:
        d_db.free()


def test_sysfs_readout_and_policy_labels_ride_with_the_runtime_block(gpu):
    """bench.gpu_sysfs(pci): best-effort power / clock read-outs of the card the library runs on (amdgpu hwmon, pp_dpm_sclk); on
    a box without the node it says why instead of raising. runtime_info() carries the labels of the unverified policies."""
    from bench import gpu_sysfs

    info = gpu.runtime_info()
    assert info["abi"] == 6
    pol = info["policies"]
    assert set(pol) >= {"comparator", "reduction", "dct"} and pol["dct"] in ("strict", "fma")
    s = gpu_sysfs(info["devices"][0]["pci"])
    assert isinstance(s, dict) and s
    if "error" not in s:
        assert any(k in s for k in ("power_w", "power_input_w", "sclk_mhz", "dpm_sclk_mhz")), s
        if "power_cap_w" in s:
            assert 100 < s["power_cap_w"] < 3000
    assert "error" in gpu_sysfs("ffff:ff:1f.7") or gpu_sysfs("ffff:ff:1f.7")  # an address that does not exist: no exception


def test_group_rearm_is_a_no_op_on_a_healthy_single_context(gpu, hvd, oracle):
    """hvd_group_rearm (ABI 6) with one context: nothing to re-arm, HVD_OK, and the library works as before; the abandoned-group
    recoveries themselves run in tests/tools/group_check.py (2, 3 and 8 contexts)."""
    lib = gpu.load()
    assert lib.hvd_group_rearm() == gpu.HVD_OK
    assert lib.hvd_group_abort() == gpu.HVD_OK and lib.hvd_group_rearm() == gpu.HVD_OK
    db, _ = hvd.synth.hash_db(5000, seed=62, plant_fraction=0.05)
    assert np.array_equal(hvd.allpairs_hamming(db, 31), oracle.allpairs(db, 31))
