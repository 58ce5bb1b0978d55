"""Data-parallel `Trainer.fit` with TWO ranks on the one-GPU box (gloo carries the collectives,
both ranks compute on device 0): unequal slabs, gradient accumulation, bucketed all-reduce,
reduce-scatter, autograd path -- tools/dp_fit_check.py holds the assertions.

This file sorts first on purpose: the two ranks are CHILD processes, and a process that has
initialised the GPU must not start programs on this pool.  It therefore runs before any test of
this process touches the GPU, and skips (instead of spawning) if something already has.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_fit_on_unequal_slabs(tmp_path):
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: children may not be started")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MRI_DIST_BACKEND="gloo",
                   MRI_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dp_fit_check.py"),
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:  # the pre-fix symptom: a rank stuck in all_reduce
            for q in procs:
                q.kill()
            pytest.fail("data-parallel fit did not finish: ranks left the collective unevenly")
        outs.append(out)
    assert [p.returncode for p in procs] == [0, 0], "\n".join(outs)
    report = json.load(open(tmp_path / "dp_fit.json"))
    assert set(report) == {"hash", "hash_rs", "siren", "batchnorm"}
    for kind, r in report.items():
        assert r["replicas_identical"] and r["finite"] and r["moved"], (kind, r)
        assert r["batches_per_epoch"] == 4 and r["steps_min"] == r["steps_max"], (kind, r)
        assert r["fused"] == (kind != "batchnorm"), (kind, r)
    assert report["hash"]["optimizer_steps"] == 4      # 2 epochs x 4 batches, 2 batches per step
    assert report["siren"]["optimizer_steps"] == 8
