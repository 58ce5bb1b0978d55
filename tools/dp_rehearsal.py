#!/usr/bin/env python3
"""Data-parallel rehearsal of the fused HIP step with real process-group reductions.

    MRI_DIST_BACKEND=gloo MRI_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 \\
        --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_rehearsal.py

(on an 8-GPU node: drop the two variables, RCCL is used).  Every rank trains the same model on
its own batches for a few steps -- level-group reductions started asynchronously, each group
Adam-stepped behind its own reduction -- then
  * all replicas must hold bitwise identical parameters, and
  * rank 0 repeats the run in ONE process on the concatenated batches (mean of equal-sized means
    = global mean): same first gradient to rounding (1e-6), same parameters to 1e-5 but for the
    few table slots where Adam amplifies rounding noise of a near-zero gradient.
`python tools/dp_rehearsal.py --single-rank-rccl` runs the same calls against a one-rank RCCL group.
Not a pytest case: a GPU test process must not start other programs.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from mri_interpolation_amd import models, parallel, trainer


def build(kind):
    torch.manual_seed(1337)
    if kind == "hash":
        net = models.HashMLP(3, 16, 2, 15, 16, 512, dim_hidden=128, n_layers=3,
                             activation=torch.nn.ReLU, batch_norm=False, final_activation=False,
                             lr=5e-3)
        with torch.no_grad():
            net.encoder.table.uniform_(-0.5, 0.5)
        return net
    return models.SirenNet(dim_in=3, dim_hidden=64, dim_out=1, n_layers=3, lr=1e-4)


def batches(rank, steps, n, lo):
    g = torch.Generator().manual_seed(100 + rank)
    return [(torch.rand(n, 3, generator=g) * (1 - lo) + lo, torch.rand(n, 1, generator=g))
            for _ in range(steps)]


def single_rank_rccl():
    """One-GPU box: RCCL refuses two ranks on one device, so the calls the data-parallel step makes
    (asynchronous in-place all-reduces of unaligned slices of the flat gradient, their waits on the
    compute stream, Adam on each slice behind its reduction) are at least run against a real
    one-rank RCCL communicator, where a sum over ranks is the identity: the step must give bit for
    bit what the same step gives without a process group."""
    import copy
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    net = build("hash").cuda()
    nets = [net, copy.deepcopy(net)]
    steps = [trainer.FusedStep(m, m.configure_optimizers(), 1) for m in nets]
    for st in steps:
        st.world, st.grad_buckets = 2, 4  # the data-parallel code path (gradients pre-divided by 2)
    real = parallel.all_reduce_async
    calls = [0]

    def through_rccl(flat):
        calls[0] += 1
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    for x, y in batches(0, 4, 20001, 0.0):
        parallel.all_reduce_async = through_rccl
        steps[0].train_step(x.cuda(), y.cuda())
        parallel.all_reduce_async = real          # no group of > 1 ranks: returns None
        steps[1].train_step(x.cuda(), y.cuda())
    parallel.barrier()
    torch.cuda.synchronize()
    assert calls[0] == 4 * 5, calls
    assert torch.equal(steps[0].flat.param, steps[1].flat.param)
    print(f"single-rank RCCL: {calls[0]} asynchronous reductions, parameters bit-identical; "
          f"rccl {torch.cuda.nccl.version()}")
    dist.destroy_process_group()


def main():
    if "--single-rank-rccl" in sys.argv:
        return single_rank_rccl()
    rank, world, local = parallel.init()
    assert world > 1, "launch with torchrun, 2 or more ranks"
    torch.cuda.set_device(local)
    steps, n = 4, 20001
    for kind, lo in (("hash", 0.0), ("siren", -1.0)):
        net = build(kind).cuda()
        step = trainer.FusedStep(net, net.configure_optimizers(), world)
        mine = batches(rank, steps, n, lo)
        grad1 = None
        for x, y in mine:
            step.train_step(x.cuda(), y.cuda())
            if grad1 is None:  # the reduced gradient of the first step (pre-divided by world)
                grad1 = step.flat.grad.detach().cpu().clone()
        torch.cuda.synchronize()
        flat = step.flat.param.detach().cpu()
        mine_flat = flat if dist.get_backend() == "gloo" else flat.cuda()  # RCCL gathers on the device
        gathered = [torch.empty_like(mine_flat) for _ in range(world)]
        dist.all_gather(gathered, mine_flat)
        assert all(torch.equal(gathered[0], t) for t in gathered), f"{kind}: replicas differ"
        if rank == 0:
            ref = build(kind).cuda()
            single = trainer.FusedStep(ref, ref.configure_optimizers(), 1)
            everyone = [batches(r, steps, n, lo) for r in range(world)]
            g_err = None
            for k in range(steps):
                x = torch.cat([everyone[r][k][0] for r in range(world)])
                y = torch.cat([everyone[r][k][1] for r in range(world)])
                single.train_step(x.cuda(), y.cuda())
                if g_err is None:
                    g = single.flat.grad.detach().cpu()
                    g_err = float((grad1 - g).abs().max() / g.abs().max())
            want = single.flat.param.detach().cpu()
            diff = (flat - want).abs() / want.abs().max()
            err = float(diff.max())
            print(f"{kind}: first-step gradient max |diff| / max |grad| = {g_err:.2e}; parameters off by "
                  f"> 1e-5: {int((diff > 1e-5).sum())} of {diff.numel()}", flush=True)
            print(f"{kind}: {world} replicas bitwise identical after {steps} steps "
                  f"({len(step._pending)} gradient groups per step); vs one process on the "
                  f"concatenated batches: max |diff| / max |param| = {err:.2e}", flush=True)
            # the first reduced gradient must agree to rounding; after a few Adam steps a handful of
            # table slots whose contributions nearly cancel (|g| ~ eps: Adam turns rounding noise of
            # g into a step of ~lr) may sit apart, the rest agrees to 1e-5
            assert g_err <= 1e-6, f"{kind}: reduced gradient differs from the single process"
            assert int((diff > 1e-5).sum()) <= diff.numel() // 1000 and err <= 2e-2, \
                f"{kind}: data-parallel parameters differ from the single process"
        parallel.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("dp rehearsal ok")


if __name__ == "__main__":
    main()
