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
    assert set(report) == {"hash", "hash_rs", "hash_plain", "hash_plain_eager", "siren", "batchnorm", "union"}
    assert report["hash_plain"]["equals_eager"]  # Trainer.fit's natively queued data-parallel steps: same bits
    # every gradient-exchange form of the fused step == one process on the concatenated batch
    union = report.pop("union")
    assert set(union) == {"all_reduce_1", "all_reduce_4", "reduce_scatter"}
    assert union["all_reduce_4"]["groups"] > 2 and union["all_reduce_1"]["groups"] == 1
    for kind, r in union.items():
        assert r["grad_rel_err"] <= 1e-6 and r["param_rel_err"] <= 1e-5, (kind, r)
    for kind, r in report.items():
        assert r["replicas_identical"] and r["finite"] and r["moved"], (kind, r)
        assert r["batches_per_epoch"] == 4 and r["steps_min"] == r["steps_max"], (kind, r)
        assert r["fused"] == (kind != "batchnorm"), (kind, r)
    assert report["hash"]["optimizer_steps"] == 4      # 2 epochs x 4 batches, 2 batches per step
    assert report["hash_plain"]["optimizer_steps"] == 8
    assert report["siren"]["optimizer_steps"] == 8


@pytest.mark.parametrize("dp_mode", ["all_reduce", "reduce_scatter", "auto"])
def test_bench_two_ranks_contract(tmp_path, dp_mode):
    """The driver's N > 1 launch of bench.py (one rank per GPU, RANK / WORLD_SIZE / MASTER_* from the
    environment), rehearsed with two ranks on the one GPU: ONE JSON line from rank 0 with the
    whole-job rate, weak scaling, and the `collectives` record a first real multi-GPU run needs."""
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: children may not be started")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MRI_DIST_BACKEND="gloo",
                   MRI_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
             "--warmup", "2", "--workload", "cfg2", "--psnr-steps", "0", "--dp-mode", dp_mode,
             "--allow-gloo"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("bench.py --gpus 2 did not finish")
        outs.append(out)
    assert [p.returncode for p in procs] == [0, 0], "\n".join(outs)
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not any(l.startswith("{") for l in outs[1].splitlines())
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 6
    assert r["config"]["global_batch"] == 2 * (1 << 18)
    c = r["collectives"]
    if dp_mode == "auto":  # what the driver's scaling pass runs: all three exchange forms, back to back
        assert set(c["legs"]) == {"all_reduce_1", "all_reduce_4", "reduce_scatter"}
        assert all(v["replicas_identical"] for v in c["legs"].values()), c["legs"]
        assert r["value"] == max(v["value"] for v in c["legs"].values())
        assert len(c["legs"]["all_reduce_4"]["groups"]) > 2 and len(c["legs"]["all_reduce_1"]["groups"]) == 1
    else:
        assert dp_mode in r["config"]["parallelism"] and len(c["legs"]) == 1
    for leg in c["legs"].values():
        assert all(g["bytes"] > 0 and g["wait_ms"] >= 0.0 for g in leg["groups"]), leg
    assert abs(r["value"] - r["config"]["global_batch"] / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["exposed_ms_per_step"] >= 0.0
    assert "all_reduce" in r["phases_ms"] and "cpu_baseline" not in r


def test_bench_one_rank_under_torchrun_is_the_plain_line(tmp_path):
    """The scaling pass's N = 1 entry (`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`, i.e. RANK=0
    WORLD_SIZE=1 in the environment) must be the same measurement as the plain `bench.py`: no process group,
    single-GPU parallelism, the natively queued step, the same workload and record format, a rate of the same
    size."""
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: children may not be started")
    lines = []
    for env_extra in ({}, dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                               MASTER_PORT=str(_free_port()))):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE") if not env_extra else ():
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40",
                            "--warmup", "10", "--psnr-steps", "0", "--no-cpu-baseline"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
        assert p.returncode == 0, p.stdout
        out = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1, p.stdout
        lines.append(json.loads(out[0]))
    plain, ranked = lines
    for key in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "records", "config", "launch"):
        assert plain[key] == ranked[key], key
    assert plain["n_gpus"] == 1 and plain["config"]["parallelism"] == "single GPU" and "collectives" not in ranked
    assert plain["launch"].startswith("one mri_fused_step call") and "packed_records" in ranked
    assert abs(plain["value"] - ranked["value"]) <= 0.15 * plain["value"], (plain["value"], ranked["value"])
