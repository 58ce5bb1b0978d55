"""N > 1 path on the CPU: two `gloo` ranks, each with its own slab batch, reduce ONE flat
gradient buffer through mri_interpolation_amd.parallel and step Adam; the result must equal
a single process stepping on the concatenated batch (mean of equal-sized local means)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import detrand
from oracle import mlp as omlp
from oracle import train as otrain


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    return otrain.HashMlpModel(3, 4, 2, 10, 4, 32, hidden=[16, 16], seed=5, table_scale=0.1)


def _batch(rank, n=256):
    x = torch.from_numpy(detrand.uniform(n * 3, 100 + rank, 0, 1).reshape(n, 3))
    # rank r samples from its own z-slab... here: slab of the slowest axis
    x[:, 0] = x[:, 0] * 0.5 + 0.5 * rank
    y = torch.from_numpy(detrand.uniform(n, 200 + rank, 0, 1).reshape(n, 1))
    return x, y


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mri_interpolation_amd import parallel
    torch.set_num_threads(1)
    r, w, _ = parallel.init(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    model = _make_model()
    opt = omlp.Adam(model.parameters(), lr=5e-3)
    for step in range(3):
        x, y = _batch(rank + 10 * step)
        _, _, grads = otrain.loss_and_grads(model, x, y)
        flat = torch.cat([g.reshape(-1) for g in grads]) / world  # pre-averaged, as the loss kernel does
        if step == 1:  # bucketed, asynchronous form used to overlap with the hash-grid backward
            cut = flat.numel() // 3
            parallel.wait_all([parallel.all_reduce_async(flat[:cut]),
                               parallel.all_reduce_async(flat[cut:])])
        else:
            parallel.all_reduce_sum(flat)
        off, synced = 0, []
        for g in grads:
            synced.append(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        opt.step(synced)
    assert parallel.all_reduce_max(float(rank), torch.device("cpu")) == world - 1
    parallel.barrier()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"),
            torch.cat([p.reshape(-1) for p in model.parameters()]).numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    np.testing.assert_array_equal(got[0], got[1])  # replicas stay bitwise identical
    model = _make_model()
    opt = omlp.Adam(model.parameters(), lr=5e-3)
    for step in range(3):
        xs, ys = zip(*[_batch(r + 10 * step) for r in range(world)])
        _, _, grads = otrain.loss_and_grads(model, torch.cat(xs), torch.cat(ys))
        opt.step(grads)
    want = torch.cat([p.reshape(-1) for p in model.parameters()]).numpy()
    err = np.abs(got[0] - want).max() / np.abs(want).max()
    assert err <= 1e-6, err


def test_env_world_defaults(monkeypatch):
    from mri_interpolation_amd import parallel
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.env_world() == (0, 1, 0)
    t = torch.ones(4)
    assert parallel.all_reduce_sum(t) is t and parallel.world_size() == 1
