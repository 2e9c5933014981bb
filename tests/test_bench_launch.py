"""bench.py must be launchable unattended at any N: `python bench.py --gpus N` spawns its own N ranks (no
torch.distributed.run, no torch import), forwards rank 0's single JSON line and returns non-zero when any
rank fails. The control channel it uses (hvd_amd.rendezvous) must not be hijackable by another local user."""
import json
import os
import socket
import stat
import struct
import subprocess
import sys
import threading
import time

import pytest

from conftest import ROOT


def _run_bench(extra_args, extra_env, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_self_launch_returns_nonzero_and_does_not_hang_when_ranks_fail():
    """Without a GPU every rank exits at once ("needs an MI355X"); the launcher must notice, stop the rest and
    return a non-zero code -- not wait in a barrier for ever, not print a JSON line."""
    import hvd_amd._lib as L

    try:
        if L.device_count() > 0:
            pytest.skip("a GPU is visible: the ranks would run")
    except Exception:
        pass
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"], {}, timeout=120)
    assert r.returncode != 0
    assert r.stdout.strip() == b""
    assert time.time() - t0 < 100


def test_rendezvous_file_is_private_and_token_is_checked(tmp_path):
    from hvd_amd import rendezvous as R

    path = str(tmp_path / "rdzv")
    box = {}

    def serve():
        try:
            box["r0"] = R.Rendezvous(0, 2, path=path, timeout=20.0)
        except Exception as exc:  # noqa: BLE001
            box["err"] = exc

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    for _ in range(200):
        if os.path.exists(path):
            break
        time.sleep(0.02)
    st = os.stat(path)
    assert stat.S_IMODE(st.st_mode) == 0o600
    port_s, token_s = open(path).read().split()
    assert len(bytes.fromhex(token_s)) == 16
    # an intruder who knows the port and the magic but not the token cannot claim rank 1
    s = socket.create_connection(("127.0.0.1", int(port_s)), timeout=5)
    s.sendall(R._MAGIC + b"\0" * 16 + struct.pack("<II", 1, 2))
    s.settimeout(5)
    assert s.recv(8) == b""  # closed without the acknowledgement
    s.close()
    assert th.is_alive()  # rank 0 is still waiting for the real rank 1
    r1 = R.Rendezvous(1, 2, path=path, timeout=20.0)
    th.join(10)
    assert "err" not in box
    got = {}
    t2 = threading.Thread(target=lambda: got.setdefault("a", box["r0"].allgather(b"zero")), daemon=True)
    t2.start()
    assert r1.allgather(b"one") == [b"zero", b"one"]
    t2.join(5)
    assert got["a"] == [b"zero", b"one"]
    r1.close()
    box["r0"].close()
    assert not os.path.exists(path)


def test_rendezvous_default_directory_is_per_user(monkeypatch, tmp_path):
    from hvd_amd import rendezvous as R

    monkeypatch.delenv("HVD_RDZV_FILE", raising=False)
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    monkeypatch.setattr(R.tempfile, "gettempdir", lambda: str(tmp_path))
    f = R._default_file()
    d = os.path.dirname(f)
    assert d == str(tmp_path / f"hvd_rdzv_{os.getuid()}")
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    os.chmod(d, 0o777)  # somebody else could now plant files: refuse
    with pytest.raises(PermissionError):
        R._default_file()


@pytest.mark.gpu
def test_bench_self_launches_two_ranks_on_one_gpu(gpu):
    """Exactly the driver's command shape with N=2 and no launcher: two ranks forced onto device 0. RCCL refuses a
    duplicate device, all ranks agree on the TCP fallback; spawn, rendezvous, exchange, JSON and exit path all run."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--hashes", "200000",
                    "--frames", "2000"],
                   {"HVD_FORCE_DEVICE": "0", "HVD_RCCL_INIT_TIMEOUT": "60"}, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["exchange"] in ("tcp-fallback", "rccl")
    assert len(out["per_rank"]) == 2
    assert len({(p["merged_pairs"], p["digest"]) for p in out["per_rank"]}) == 1  # both ranks hold the same pair list
    assert all(p["kernel_ms"] > 0 for p in out["per_rank"])
    assert sum(p["pairs"] for p in out["per_rank"]) == out["config"]["pairs_found"]
    assert "scale_metric" in out and out["scaling"] == "weak"
    _check_breakdown(out, steps=2, world=2)
    # the prediction the first real N > 1 line is to be held against rides in the line itself
    pred = out["predicted"]
    assert pred["n_gpus"] == 2 and pred["n_hashes"] == out["config"]["n_hashes"] and pred["ms_per_step"] > 0
    assert 0.5 < pred["efficiency"] <= 1.0 and 0.99 < pred["imbalance"] < 1.05 and pred["measured_over_predicted"] > 0
    assert b"import torch" not in open(os.path.join(ROOT, "bench.py"), "rb").read()


@pytest.mark.gpu
def test_bench_self_launch_propagates_a_rank_failure(gpu):
    """One rank dies (bad device index) -> the launcher returns non-zero promptly and prints no JSON line."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--hashes", "100000"],
                   {"HVD_FORCE_DEVICE": "97"}, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == b""


@pytest.mark.gpu
def test_single_process_bench_runs_two_contexts_on_one_gpu():
    """`python bench.py --gpus 2 --single-process --devices 0,0`: the two ranks are the two contexts of the library's
    in-process device group (the mode the drop-in surfaces use under HVD_DEVICES), one thread each; device 0 listed twice,
    so the candidates meet in host memory. One JSON line, exit code 0, the riding config-4 / config-5 legs included."""
    r = _run_bench(["--gpus", "2", "--single-process", "--devices", "0,0", "--steps", "2", "--warmup", "1", "--hashes", "200000",
                    "--cfg5-videos", "2000", "--no-cpu-baseline"], {}, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["launch"].startswith("single process")
    assert out["config"]["exchange"].startswith("host-memory")
    assert len(out["per_rank"]) == 2 and all(p["pairs"] > 0 for p in out["per_rank"])
    assert out["config5"]["planted_recall"] == 1.0 and out["config4"]["pairs_found"] > 0
    _check_breakdown(out, steps=2, world=2)
    assert out["runtime"]["group_exchange"] == "host"
    # both scaling curves from one run: the weak headline and BASELINE configs[2] itself (fixed total work)
    assert out["strong"]["scaling"] == "strong" and out["strong"]["n_gpus"] == 2 and out["strong"]["value"] > 0
    assert len(out["strong"]["per_rank"]) == 2 and out["strong"]["per_rank"][0]["steps"]["kernel_ms"]
    # config 4 / config 5 carry the same split
    assert len(out["config4"]["per_rank"]) == 2 and "exchange_ms" in out["config4"]["per_rank"][1]
    st = out["config5"]["stages_ms"]
    for key in ("hash_ms", "gather_ms", "compact_ms", "search_ms", "search_local_ms", "search_exchange_ms", "search_fold_ms"):
        assert st[key] >= 0, key
    assert st["search_local_ms"] > 0 and st["search_exchange_ms"] > 0 and st["gather_ms"] > 0
    assert [p["rank"] for p in out["config5"]["per_rank"]] == [0, 1]


def _check_breakdown(out, steps, world):
    """every rank's per-step split and what the ranks run on are in the one JSON line."""
    keys = ("expand_ms", "kernel_ms", "readback_ms", "exchange_ms", "host_ms", "step_ms")
    assert [p["rank"] for p in out["per_rank"]] == list(range(world))
    for p in out["per_rank"]:
        for k in keys:
            assert len(p["steps"][k]) == steps and isinstance(p[k], float), (k, p)
        assert all(x > 0 for x in p["steps"]["kernel_ms"]) and all(x > 0 for x in p["steps"]["exchange_ms"])
        for i in range(steps):  # the parts add up to the step
            parts = sum(p["steps"][k][i] for k in keys[:-1])
            assert abs(parts - p["steps"]["step_ms"][i]) < 0.02, (parts, p["steps"]["step_ms"][i])
    rt = out["runtime"]
    assert rt["rccl_version"] > 20000 and rt["hip_runtime_version"] > 0 and "librccl" in rt["librccl_path"]
    assert rt["devices"] and "gfx950" in rt["devices"][0]["arch"]
    assert [r["rank"] for r in rt["ranks"]] == list(range(world)) and all("device" in r and "pci" in r for r in rt["ranks"])
    assert rt["xgmi_or_pcie"] and rt["exchange"] == out["config"]["exchange"]
