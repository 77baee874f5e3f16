"""The RCCL path executed on ONE MI355X: a child process with WORLD_SIZE=1 and backend "nccl" runs KD steps with the
bucketed reducer forced on (real asynchronous `all_reduce` calls on the flat gradient buffer through ProcessGroupNCCL /
RCCL) and must reproduce the reducer-less step bit for bit; buckets fire head -> fusion/FPN/LiDAR -> camera; the
reducer adds no host synchronisation; the step also replays from a hipGraph with the collectives captured inside."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    out = tmp_path_factory.mktemp("rccl") / "res.json"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()),
               KD_RCCL_OUT=str(out), OMP_NUM_THREADS="2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_world1_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


def test_rccl_communicator_comes_up(result):
    assert result["backend"] == "nccl" and result["world"] == 1 and result["ranks_seen"] == 1


@pytest.mark.parametrize("fusion", ("weighted", "minimal"))
def test_forced_reducer_steps_are_bit_identical_and_ordered(result, fusion):
    r = result[fusion]
    assert r["bit_identical_steps"] == [True, True, True], r         # parameters, gradients, Adam moments, BN buffers
    assert r["orders"] == [[2, 1, 0]] * 3, r["orders"]               # bucket launch order = backward completion order
    assert r["collectives"] == 9, r                                   # 3 buckets x 3 steps really went to RCCL
    assert r["host_syncs_forced"] <= r["host_syncs_plain"], r         # the reducer adds no host synchronisation
    assert set(r["sync_sites_forced"]) <= set(r["sync_sites_plain"]), r


def test_step_with_collectives_replays_from_a_hipgraph(result):
    g = result["graph"]
    assert g["ok"], g
    assert g["bit_identical"], g
