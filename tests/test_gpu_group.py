"""The in-process device group (hvd_init_devices / HVD_DEVICES): the drop-in entry points shard over every listed GPU
inside ONE process, as the single-process reference needs (entrypoint.py:235 -> dedup.py:213). The group is process state,
so the checks run in a process of their own (tests/tools/group_check.py, against the CPU oracle). One GPU listed twice is
what a 1-GPU box can exercise: two contexts on two streams, tile (rb + cb) % 2, exchange through host memory; on a node
with more GPUs the second test runs the same checks over RCCL."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(devs, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("HVD_DEVICES", "HVD_DEVICE")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "group_check.py"), devs], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0 and "GROUP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_group_of_one_device_listed_twice(gpu):
    assert "exchange host" in _run("0,0")


def test_group_of_three_contexts_on_one_device(gpu):
    _run("0,0,0")


def test_group_of_eight_contexts_on_one_device(gpu):
    """World 8 -- the size of the driver's scale run -- through every drop-in surface, against the oracle: the tile partition
    (rb + cb) % 8, the fixed-slot candidate exchange and the key-list exchange of the video search with eight participants."""
    _run("0,0,0,0,0,0,0,0")


def test_group_over_rccl_when_the_node_has_more_gpus(gpu):
    n = gpu.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL refuses a duplicate device (the host-memory exchange is tested above)")
    assert "exchange rccl" in _run(",".join(str(d) for d in range(min(n, 8))))


def test_hvd_devices_environment_turns_the_default_init_into_a_group(gpu):
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, hvd_amd; from hvd_amd import _lib as L, synth; "
            "from oracle import oracle as O; O.build(); L.ensure(); assert L.context_count() == 2, L.context_count(); "
            "db, _ = synth.hash_db(20000, seed=5, plant_fraction=0.02); "
            "assert np.array_equal(hvd_amd.allpairs_hamming(db, 31), O.allpairs(db, 31, num_threads=8)); print('ENV_OK')" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "HVD_DEVICE"}
    env["HVD_DEVICES"] = "0,0"
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, text=True)
    assert r.returncode == 0 and "ENV_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
