#!/usr/bin/env python3
"""One rank of a data-parallel `Trainer.fit` / `launcher.main` check (started by
tests/test_00_dp_two_rank_gpu.py, or by torchrun on a multi-GPU node).

    MRI_DIST_BACKEND=gloo MRI_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 \\
        --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dp_fit_check.py OUT

The volume's slow axis (25 slices) does not divide by the world size, so the ranks' slabs differ
in size: before round 2 the ranks ran 4 and 3 batches per epoch and the run hung in the gradient
all-reduce.  Checks, per model family (hash + tiny MLP on the bucketed all-reduce path and on the
reduce-scatter path, SIREN on the flat all-reduce path, the BatchNorm decoder on the autograd
path): every rank runs the same number of steps, and the replicas hold bitwise identical
parameters after `fit`.  Then every fused gradient-exchange form (plain all-reduce, bucketed
all-reduce, reduce-scatter) against ONE process stepping on the concatenated batch
(`union_check`).  Rank 0 writes OUT/dp_fit.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from mri_interpolation_amd import config as cfg, datamodules, models, parallel
from mri_interpolation_amd.trainer import Trainer


def build(kind):
    torch.manual_seed(1337)
    if kind.startswith("hash"):
        return models.HashMLP(3, 8, 2, 14, 8, 64, dim_hidden=64, n_layers=3,
                              activation=torch.nn.ReLU, batch_norm=False, final_activation=False,
                              lr=5e-3)
    if kind == "siren":
        return models.SirenNet(dim_in=3, dim_hidden=64, dim_out=1, n_layers=3, lr=1e-4)
    return models.HashMLP(3, 4, 2, 12, (8, 8, 8), (32, 32, 32), dim_hidden=32, n_layers=2,
                          lr=5e-3)  # the reference's BatchNorm + GELU decoder: autograd path


def union_check(rank, world):
    """Each data-parallel form of the fused step against a single process on the union batch: the mean
    over equal-sized local means is the global mean, so the reduced gradient of the first step must
    agree to rounding and so must the parameters after it wherever Adam is well conditioned."""
    from mri_interpolation_amd.trainer import FusedStep
    n, out = 4096, {}

    def batch(r):
        g = torch.Generator().manual_seed(100 + r)
        return torch.rand(n, 3, generator=g).cuda(), torch.rand(n, 1, generator=g).cuda()

    ref = build("hash").cuda()
    one = FusedStep(ref, ref.configure_optimizers(), 1)
    one.train_step(torch.cat([batch(r)[0] for r in range(world)]),
                   torch.cat([batch(r)[1] for r in range(world)]))
    torch.cuda.synchronize()
    g_one, p_one = one.flat._grad_all.clone(), one.flat._param_all.clone()
    for name, mode, buckets in (("all_reduce_1", "all_reduce", 1), ("all_reduce_4", "all_reduce", 4),
                                ("reduce_scatter", "reduce_scatter", 1)):
        net = build("hash").cuda()
        step = FusedStep(net, net.configure_optimizers(), world)
        step.dp_mode, step.grad_buckets = mode, buckets
        step.train_step(*batch(rank))
        torch.cuda.synchronize()
        lo, hi = (0, step.flat.numel) if mode == "all_reduce" else \
            parallel.shard_range(step.flat._grad_all.numel(), rank, world)  # the reduced part of the buffer
        g = step.flat._grad_all[lo:hi]
        g_err = float((g - g_one[lo:hi]).abs().max() / g_one.abs().max())
        safe = g_one.abs() > 1e-3 * g_one.abs().max()  # |g| >> eps: Adam's step does not amplify rounding
        p_err = float(((step.flat._param_all - p_one).abs() * safe).max() / p_one.abs().max())
        worst = torch.tensor([g_err, p_err], dtype=torch.float64)
        if dist.get_backend() != "gloo":
            worst = worst.cuda()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        out[name] = dict(grad_rel_err=float(worst[0]), param_rel_err=float(worst[1]),
                         groups=len(step.last_group_bytes))
        assert float(worst[0]) <= 1e-6, f"{name}: reduced gradient differs from the single process: {out[name]}"
        assert float(worst[1]) <= 1e-5, f"{name}: parameters differ from the single process: {out[name]}"
    return out


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
    rank, world, local = parallel.init()
    assert world > 1, "launch with 2 or more ranks"
    torch.cuda.set_device(local)
    vol = datamodules.phantom_volume((25, 32, 32)).cpu().numpy()
    report = {}
    finals = {}
    for kind in ("hash", "hash_rs", "hash_plain", "hash_plain_eager", "siren", "batchnorm"):
        c = (cfg.HashConfig() if kind != "siren" else cfg.BaseConfig()).resolve(vol.shape)
        c.batch_size = 4096
        dm = datamodules.MriDataModule(config=c, volume=vol, norm_siren=kind == "siren")
        dm.prepare_data()
        loader = dm.train_dataloader(rank, world)
        net = build(kind).cuda()
        # "hash": the bucketed, overlapped form with accumulation; "hash_plain": ONE all-reduce per step, which
        # Trainer.fit queues through SteadyLoop (mri_fused_step, reduce, Adam) -- "hash_plain_eager" is the same
        # run queued op by op and must end on the same bits
        tr = Trainer(max_epochs=2, accumulate_grad_batches=2 if kind == "hash" else None,
                     dp_mode="reduce_scatter" if kind == "hash_rs" else "all_reduce",
                     grad_buckets=4 if kind == "hash" else 1, native_steps=kind != "hash_plain_eager")
        tr.fit(net, loader)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        mine = flat.cpu() if dist.get_backend() == "gloo" else flat
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        steps = torch.tensor([float(tr.global_step)])
        if dist.get_backend() != "gloo":
            steps = steps.cuda()
        hi, lo = steps.clone(), steps.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        report[kind] = dict(batches_per_epoch=len(loader), optimizer_steps=tr.global_step,
                            steps_min=int(lo), steps_max=int(hi), replicas_identical=bool(same),
                            fused=tr.fused is not None, finite=bool(torch.isfinite(flat).all()),
                            moved=bool((flat != torch.cat([p.detach().reshape(-1) for p in
                                                           build(kind).cuda().parameters()])).any()))
        finals[kind] = flat.clone()
        assert same, f"{kind}: replicas differ after fit"
        assert int(lo) == int(hi), f"{kind}: ranks ran {int(lo)}..{int(hi)} steps"
    assert torch.equal(finals["hash_plain"], finals["hash_plain_eager"]), "natively queued data-parallel fit differs"
    report["hash_plain"]["equals_eager"] = True
    report["union"] = union_check(rank, world)
    parallel.barrier()
    dist.destroy_process_group()
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "dp_fit.json"), "w") as f:
            json.dump(report, f)
        print("dp fit check ok", json.dumps(report))


if __name__ == "__main__":
    main()
