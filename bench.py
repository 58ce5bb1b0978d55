#!/usr/bin/env python3
"""Headline benchmark: coordinate-samples/sec of one full training step (on-device batch
generation -> forward -> MSE -> backward -> [RCCL all-reduce] -> Adam) on a synthetic volume.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg2|cfg3]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`:
one rank per GPU, z-slab sharding of the volume, equal local batch (weak scaling), ONE
all-reduce of the flat gradient buffer per step.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs; the metric is quoted on a 256^3 volume at 1/2/4/8 GPUs):
  cfg4  256^3, hash L16 F2 T2^19 base 16 growth 1.4 + ReLU MLP 32-128-128-1, B=2^18 per GPU  (default)
  cfg2  128^3, hash L16 F2 T2^19 16->512 + ReLU MLP 32-64-64-1, B=2^18
  cfg3  256^3, SIREN 3-256x5-1, coords in [-1,1], B=2^20
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL shares device buffers between the ranks of a node through dmabuf IPC; the variable is read
# when HIP initialises, so it is set before torch is imported -- for a multi-process launch only, and an
# explicit setting wins (mri_interpolation_amd/parallel.py: ipc_default)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # exact-f32 matrix rate (no xf32 on gfx950)
# The 128-wide decoder and the SIREN chain multiply on the bf16 pipe with operands split exactly into
# three bf16 terms: six bf16 MFMAs per f32-accurate product (csrc/bf16x3.h).  Their ceiling in
# ALGORITHMIC (f32) flops is the dense bf16 peak (16 x the f32 MFMA rate, MI355X_MICROARCH.md) / 6.
MFMA_BF16_PEAK_TF = 16 * MFMA_F32_PEAK_TF
MFMA_X3_PEAK_TF = MFMA_BF16_PEAK_TF / 6
# what "dtype": "f32" means here, said in the line itself: every value is f32 and every result is within
# f32 rounding of an f32 evaluation; the matrix products reach that on the bf16 pipe
ARITHMETIC = ("f32 values and f32 accumulation everywhere; matrix products as six bf16 MFMAs per product on "
              "operands split EXACTLY into three bf16 terms (per-product error below f32 rounding, measured "
              "equal to v_mfma_f32_32x32x2_f32: DESIGN.md 4.4, tests/test_bf16x3_cpu.py); hash / gather / "
              "Adam in plain f32; table gradient: ")
# the table-gradient records (mri_set_option("bwd_records")): what is added up, said in the line
RECORDS = {
    "f32": (0, "every contribution w*g is the f32 product the reference's autograd forms "
               "(encoding.py:127-128), the sum over a slot is EXACT (64-bit fixed point, bitwise reproducible) "
               "and rounded once to f32"),
    "packed": (1, "each contribution w*g rounded to 18-21 significant bits (8-byte packed records) before an "
                  "exact integer sum: NOT f32-equivalent, 4-64x the rounding of an f32 product per contribution "
                  "(tests/test_gpu_round3.py::test_record_formats_per_slot_against_float64)"),
}


def arithmetic(records):
    return ARITHMETIC + RECORDS[records][1]


def mfma_roof(step, achieved_tf):
    """Roofline entry of a matrix-bound phase: which pipe the model's products run on."""
    x3 = bool(getattr(step, "use_chain", False)) or (
        bool(getattr(step, "use_tiny", False)) and step.layers[0].weight.shape[0] in (64, 128)
        and step.layers[0].weight.shape[1] <= 32)
    peak = MFMA_X3_PEAK_TF if x3 else MFMA_F32_PEAK_TF
    roof = dict(bound="mfma", achieved=achieved_tf, peak=peak, unit="TFLOP/s",
                pipe=("bf16 MFMA, 6 products of exact three-term operands per f32 product: "
                      "peak = 2517 TF bf16 dense / 6") if x3 else "f32 MFMA")
    if x3:
        roof["vs_f32_mfma_peak"] = achieved_tf / MFMA_F32_PEAK_TF
    return roof

WORKLOADS = {
    "cfg4": dict(shape=(256, 256, 256), model="hash", finest=16 * 1.4 ** 15, hidden=128,
                 batch=1 << 18, lr=5e-3, norm_siren=False),
    "cfg2": dict(shape=(128, 128, 128), model="hash", finest=512, hidden=64, batch=1 << 18,
                 lr=5e-3, norm_siren=False),
    "cfg3": dict(shape=(256, 256, 256), model="siren", hidden=256, batch=1 << 20, lr=1e-4,
                 norm_siren=True),
    # BASELINE config 5 on its real workload: the reference's sample volume (352 x 352 x 6 x 15,
    # tests/golden/sample_volume.npz -- int16 voxels + scl_slope, a data fixture), 4-D encoder
    # (16 corners per level) and the held-out-frame protocol: the even time frames are trained
    # on (coordinates from the full t grid, reference interp.py:27,35), PSNR is reported on the 7
    # held-out odd frames next to linear interpolation in t (interp.py's baseline).
    # Per-axis resolutions (the reference's MultiResHashGridV2, selected by tuple arguments as in
    # models.py:691-708): x, y grow as in config 4, z and t keep one grid node per slice / per
    # TRAINED frame on every level, so features are interpolated linearly between trained frames.
    # (The isotropic grid of config 4 puts many untrained cells between two frames: 62 dB on the
    # trained frames but 16 dB on the held-out ones, against 44 dB for linear interpolation.)
    "cfg5": dict(shape=(352, 352, 6, 15), model="hash", base=(16, 16, 5, 7),
                 finest=(16 * 1.4 ** 15, 16 * 1.4 ** 15, 5, 7), hidden=128,
                 batch=1 << 18, lr=5e-3, norm_siren=False, holdout=True, sample_volume=True),
}


def load_volume(w, dev):
    """The workload's volume in HBM: the analytic phantom, or -- config 5 -- the sample volume."""
    import numpy as np
    import torch
    from mri_interpolation_amd import datamodules
    if not w.get("sample_volume"):
        return datamodules.phantom_volume(w["shape"], device=dev), "synthetic"
    z = np.load(os.path.join(ROOT, "tests", "golden", "sample_volume.npz"))
    meta = json.loads(str(z["meta"]))
    vol = (z["raw_int16"].astype(np.float64) * meta["scl_slope"] + meta["scl_inter"])
    assert tuple(vol.shape) == tuple(w["shape"])
    return torch.from_numpy(vol.astype(np.float32)).to(dev), \
        "sample_ankle_dyn_mri.nii.gz (the reference's sample volume, committed as a data fixture)"


def build_model(w):
    import torch
    from mri_interpolation_amd import models
    torch.manual_seed(1337)  # reference launcher.py:30
    if w["model"] == "hash":
        return models.HashMLP(dim_in=len(w["shape"]), n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                              base_resolution=w.get("base", 16), finest_resolution=w["finest"],
                              dim_hidden=w["hidden"], dim_out=1, n_layers=3,
                              activation=torch.nn.ReLU, batch_norm=False, final_activation=False,
                              lr=w["lr"])
    return models.SirenNet(dim_in=3, dim_hidden=w["hidden"], dim_out=1, n_layers=5, lr=w["lr"])


def phase_model(w, step, n_params):
    """Algorithmic bytes / flops per step of each timed phase."""
    b = w["batch"]
    dims = [l.weight.shape[1] for l in step.layers] + [step.layers[-1].weight.shape[0]]
    mac = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    out = {"mlp_fwd": ("mfma", 2.0 * mac * b), "mlp_bwd": ("mfma", 4.0 * mac * b),
           "mlp_fused": ("mfma", 6.0 * mac * b), "adam": ("hbm", 28.0 * n_params)}
    if step.encoder is not None:
        e = step.encoder
        corners = (1 << e.dim) * e.n_levels * e.n_features_per_level * 4
        per_coord = 4 * e.dim + corners + e.output_dim * 4  # coords + table rows + features
        out["hashgrid_fwd"] = ("hbm", float(per_coord) * b)
        out["hashgrid_bwd"] = ("hbm", float(per_coord) * b)
        if getattr(step, "fuse_table_adam", False) and step.world == 1:
            # the table's Adam step rides on the last stage: parameter and two moments read and
            # written there (24 B per table parameter), the gradient itself stays on chip
            n_table = e.table.numel()
            out["hashgrid_bwd"] = ("hbm", float(per_coord) * b + 24.0 * n_table)
            out["adam"] = ("hbm", 28.0 * (n_params - n_table))
    return out


def pmc_traffic(workload, phase, records="f32"):
    """HBM bytes per launch of the phase's kernels from the committed rocprofv3 PMC summary
    (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections applied as the summary
    states) -- measured offline by tools/summarize_prof.py, None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    entry = table.get(workload, {})
    if records != "f32":  # summaries of the other record format are filed under "<workload>_<records>"
        entry = table.get(f"{workload}_{records}", {})
    return entry.get(phase)


def cpu_baseline(w, name):
    """The oracle (CPU restatement of the reference's step) timed on the host cores, on a
    bounded sample of the same workload."""
    import torch
    from oracle import train as otrain
    # the threads this process may run on (the box gives one GPU a share of the host's cores:
    # the affinity mask says which; os.cpu_count() reports the whole host), 16 at most
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cores = int(os.environ.get("MRI_CPU_THREADS", max(1, min(allowed, os.cpu_count() or 1, 16))))
    torch.set_num_threads(cores)
    b = 1 << 15 if w["model"] == "hash" else 1 << 14
    dim = len(w["shape"])
    if w["model"] == "hash":
        m = otrain.HashMlpModel(dim, 16, 2, 19, w.get("base", 16), w["finest"],
                                hidden=[w["hidden"]] * 2, seed=1)
    else:
        m = otrain.SirenModel(dim, w["hidden"], 1, 5, seed=1)
    g = torch.Generator().manual_seed(0)
    lo = -1.0 if w["norm_siren"] else 0.0
    batches = [(torch.rand(b, dim, generator=g) * (1 - lo) + lo, torch.rand(b, 1, generator=g))
               for _ in range(2)]
    opt = None
    _, opt = otrain.train_steps(m, batches[:1], w["lr"], opt)  # warm-up
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < 12.0 and steps < 40:  # ~12 s of host work
        _, opt = otrain.train_steps(m, [batches[steps % 2]], w["lr"], opt)
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=b * steps / dt, unit="coord-samples/s", cores=cores, kind="port",
                sample=f"{steps} oracle train steps (PyTorch-CPU restatement of the reference "
                       f"step) of {name} at batch {b} instead of {w['batch']} (a full-size step takes the "
                       f"host several seconds: the sample is bounded to ~12 s)")


def run_predict(args, w, vol, step, model, rank, world, dev, data_name):
    """Inference throughput of launcher.py's predict / interpolate passes (reference
    launcher.py:179-222): ordered dense-grid batches of 2^20 coordinates, generated on device,
    through the forward kernels only.  One step = one batch; rank r predicts its own slab."""
    import torch
    from mri_interpolation_amd import datamodules, parallel
    batch = 1 << 20
    full = datamodules.MriImage(volume=vol, norm_siren=w["norm_siren"], device=dev)
    lo, hi = parallel.voxel_range(full.shape, rank, world)
    loader = datamodules.DeviceLoader(full, batch, shuffle=False, lo=lo, hi=hi, drop_last=True)
    per_epoch = len(loader)
    assert per_epoch > 0, "volume smaller than one predict batch"
    idx = torch.empty(batch, dtype=torch.int64, device=dev)
    coords = torch.empty(batch, full.dim_in, device=dev)
    events, k = {}, [0]

    def one_step(sample):
        b = k[0] % per_epoch
        k[0] += 1
        first, n = loader.span(b)
        step.phase_events = events if sample else None
        with step._phase("coords"):
            full.batch(loader.indices(first, n, out=idx), coords, None)
        with torch.no_grad():
            return step.forward(coords, train=False)[0]

    for _ in range(args.warmup):
        one_step(False)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pred = one_step(i % max(1, args.phase_every) == 0)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.all_reduce_max(time.perf_counter() - t0, dev)
    step.phase_events = events
    phases = step.phase_ms()
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    dims = [l.weight.shape[1] for l in step.layers] + [step.layers[-1].weight.shape[0]]
    mac = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    pm = {"mlp_fwd": ("mfma", 2.0 * mac * batch)}
    if step.encoder is not None:
        e = step.encoder  # SURVEY.md 8(d), inference: 4 D + L 2^D F 4 (+ 4 for the intensity)
        pm["hashgrid_fwd"] = ("hbm", float(4 * e.dim + (1 << e.dim) * e.output_dim * 4
                                           + e.output_dim * 4) * batch)
    dominant = max((p for p in phases if p in pm), key=lambda p: phases[p])
    bound, amount = pm[dominant]
    sec = phases[dominant] * 1e-3
    if bound == "hbm":
        roof = dict(bound="hbm", achieved=amount / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
    else:
        roof = mfma_roof(step, amount / sec / 1e12)
    roof.update(frac=roof["achieved"] / roof["peak"],
                traffic=pmc_traffic(args.workload + "_predict", dominant), kernel=dominant,
                ms_per_launch=phases[dominant])
    value = batch * world * args.steps / elapsed
    result = {"metric": "coord-samples/sec (predict)", "value": value, "unit": "coord-samples/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "arithmetic": arithmetic(args.records),
              "data": data_name,
              "config": {"workload": f"{args.workload} predict: dense grid "
                                     f"{'x'.join(map(str, full.shape))}, {w['model']}, "
                                     f"batch {batch} coords per GPU and step",
                         "global_batch": batch * world,
                         "params": sum(p.numel() for p in model.parameters()),
                         "parallelism": f"dp{world} z-slab" if world > 1 else "single GPU"},
              "roofline": roof, "phases_ms": {p: round(v, 4) for p, v in sorted(phases.items())},
              "phases_sampled_every": max(1, args.phase_every),
              "checksum": float(pred.double().sum())}
    if step.encoder is not None:
        e = step.encoder
        per_coord = 4 * e.dim + (1 << e.dim) * e.output_dim * 4 + 4
        result["step_hbm_frac"] = value / world * per_coord / (HBM_PEAK_GBS * 1e9)
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--psnr-steps", type=int, default=None,
                    help="total training steps before the PSNR evaluation (0 = skip; default 2000, "
                         "SIREN workloads 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-group", type=int, default=1,
                    help="batches produced per launch pair of the on-device producer (BatchPipeline.group): the "
                         "same batches bit for bit, produced once in `group` steps instead of beside every lookup "
                         "(measured: the lookup gets its 8 us back, the step does not -- the work only moves)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="produce every batch on the main stream at the start of its step")
    ap.add_argument("--cold-start", action="store_true",
                    help="time the first leg on a device that has just left idle (the order until round 3: timed legs, "
                         "then the quality leg); by default the quality leg's training steps run first, so that every "
                         "timed leg sees settled clocks")
    ap.add_argument("--phase-every", type=int, default=0,
                    help="bracket the phases with HIP events on every n-th timed step only (the 4th, the (n+4)th, "
                         "...): a timing event is a serialisation point, five per step cost 0.15 ms (0.71 against "
                         "0.56 ms per step with n = 1, round 3).  0 (default) = max(6, steps // 5): five sampled "
                         "steps in a long run, three in the driver's 20-step form (steps 4, 10, 16), so that "
                         "phases_ms / rooflines.*.ms_per_launch are means of >= 3 launches")
    ap.add_argument("--launch", default="native", choices=["native", "eager"],
                    help="how a step is queued (trainer.SteadyLoop): native = one mri_fused_step call per step; "
                         "eager = op by op from Python")
    ap.add_argument("--fixed-batch", action="store_true",
                    help="diagnostic: train on the first batch over and over (no batch is produced inside the "
                         "timed steps): what the on-device batch producer costs a step; never a measured line")
    ap.add_argument("--bwd-method", type=int, default=0)
    ap.add_argument("--split", type=float, default=None,
                    help="fraction of the batch in the first slice of the fused decoder step "
                         "(0 = one slice); default: FusedStep's")
    ap.add_argument("--opt", action="append", default=[],
                    help="library tuning option name=value (mri_set_option), repeatable")
    ap.add_argument("--overlap-forward", action="store_true",
                    help="run the hash-grid lookup beside the decoder kernel (FusedStep.overlap_forward)")
    ap.add_argument("--no-count-ahead", action="store_true",
                    help="count the table-gradient records inside their own step (FusedStep.count_ahead = False)")
    ap.add_argument("--count-ahead", action="store_true",
                    help="count them one step ahead whatever the decoder (default: the 128-wide decoder only)")
    ap.add_argument("--fused-adam", action="store_true",
                    help="apply the table's Adam step where its gradient is complete "
                         "(FusedStep.fuse_table_adam; measured -2 %% on cfg4, +2 %% on cfg2 / cfg5: off)")
    ap.add_argument("--mode", default="train", choices=["train", "predict"],
                    help="predict: inference throughput of the same model on the dense grid of "
                         "the volume (launcher.py's predict / interpolate passes), no training")
    ap.add_argument("--dp-mode", default="auto", choices=["auto", "all_reduce", "reduce_scatter"],
                    help="gradient exchange of the data-parallel step (FusedStep.dp_mode); auto (several "
                         "GPUs): plain all-reduce, bucketed all-reduce and reduce-scatter legs back to back, "
                         "`value` = the fastest leg whose replicas stayed identical")
    ap.add_argument("--leg-timeout", type=float, default=240.0,
                    help="seconds the legs after the first may take before the line of the finished legs is printed")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="rehearsal: accept a gloo process group for --gpus > 1 (never a measured line)")
    ap.add_argument("--records", default="f32", choices=sorted(RECORDS),
                    help="table-gradient records of the headline leg (mri_set_option bwd_records): f32 = the "
                         "reference's precision; packed = 8-byte records, each contribution rounded to 18-21 bits")
    ap.add_argument("--no-records-leg", action="store_true",
                    help="skip the second timed leg with the other record format")
    ap.add_argument("--grad-buckets", type=int, default=0,
                    help="level groups of the table-gradient kernels (0 = default)")
    args = ap.parse_args()
    if args.phase_every <= 0:
        args.phase_every = max(6, args.steps // 5)

    import torch
    from mri_interpolation_amd import _lib, datamodules, parallel, trainer
    _lib.load()  # no fallback: fail before touching the GPU if the HIP library is missing

    _lib.set_option("bwd_records", RECORDS[args.records][0])
    for item in args.opt:  # (after --records: an explicit bwd_records=2 selects round 1's layout for A/B runs)
        name, value = item.split("=")
        _lib.set_option(name, int(value))

    rank, world, local = parallel.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun")
    dist_info = None
    if world > 1:
        import torch.distributed as dist
        dist_info = dict(backend=dist.get_backend(), ranks=dist.get_world_size(),
                         rccl=".".join(map(str, torch.cuda.nccl.version()))
                         if dist.get_backend() == "nccl" else None)
        # a multi-GPU line must be an RCCL line with every rank present: anything else fails HERE
        if dist_info["ranks"] != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist_info['ranks']} ranks")
        if dist_info["backend"] != "nccl" and not args.allow_gloo:
            raise SystemExit(f"--gpus {args.gpus}: backend is '{dist_info['backend']}', not nccl (RCCL); "
                             "a CPU / one-GPU rehearsal must say --allow-gloo")
        if dist_info["backend"] == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    w = WORKLOADS[args.workload]
    if args.psnr_steps is None:  # 2,000 steps (1.3 s of hash steps; SIREN: 200 = 4 s)
        args.psnr_steps = 2000 if w["model"] == "hash" else 200

    # synthetic volume in HBM, this rank's z-slab, on-device batch generation
    vol, data_name = load_volume(w, dev)
    ds = datamodules.MriImage(volume=vol, norm_siren=w["norm_siren"], device=dev,
                              frames=slice(0, None, 2) if w.get("holdout") else None)
    lo, hi = parallel.voxel_range(ds.shape, rank, world)
    # several ranks: the same number of full batches on every rank (slabs may differ in size)
    loader = datamodules.DeviceLoader(ds, w["batch"], shuffle=True, lo=lo, hi=hi, drop_last=True,
                                      seed=1337 + rank,
                                      steps=datamodules.sharded_steps(ds.shape, w["batch"], world)
                                      if world > 1 else None)
    model = build_model(w).to(dev)
    opt = model.configure_optimizers()
    step = trainer.FusedStep(model, opt, world)
    step.bwd_method = args.bwd_method
    if args.dp_mode != "auto":
        step.dp_mode = args.dp_mode
    step.overlap_forward = args.overlap_forward
    step.fuse_table_adam = args.fused_adam
    if args.no_count_ahead:
        step.count_ahead = False
    if args.count_ahead:
        step.count_ahead = True
    if args.grad_buckets:
        step.grad_buckets = args.grad_buckets
    if args.split is not None:
        step.split_fraction = args.split
    n_params = sum(p.numel() for p in model.parameters())
    if args.mode == "predict":
        return run_predict(args, w, vol, step, model, rank, world, dev, data_name)
    counter = [0]
    events = {}
    sampling = [False]
    leg_start = [0]
    # shuffled batches, epoch after epoch; batch k+1 is produced while step k runs (queued on
    # the step's side stream, or after Adam when the step has none): its kernels still execute
    # inside the timed region
    pipe = datamodules.BatchPipeline(loader, group=1 if args.no_prefetch else args.batch_group)

    # One GPU, fused hash-grid + tiny-MLP step: queued by ONE library call per step (trainer.SteadyLoop ->
    # mri_fused_step: the same launches on the same data, bit-identical parameters) -- queued op by op from
    # Python a step costs 0.2-0.55 ms of host time depending on the box, against 0.52 ms on the GPU.  The
    # event-bracketed sample steps (every --phase-every-th) run eagerly, as do all steps with --launch eager.
    graphed, launch_note = [None], [None]

    def capture_graphs():
        graphed[0] = None
        if args.launch == "eager" or args.fixed_batch or args.no_prefetch:
            return
        if trainer.SteadyLoop.unsupported(step, pipe) is None:
            try:
                graphed[0] = trainer.SteadyLoop(step, pipe, mode=args.launch).capture()
            except (RuntimeError, ValueError) as exc:  # never lose the line to the launch form: queue eagerly
                graphed[0], launch_note[0] = None, f"eager (the {args.launch} form was refused: {exc})"
            counter[0] = pipe.k  # (its warm-up steps count as steps of the run)

    def one_step():
        k = counter[0]
        counter[0] += 1
        every = max(1, args.phase_every)
        step.phase_events = events if sampling[0] and (k - leg_start[0]) % every == min(3, every - 1) else None
        if graphed[0] is not None:
            return graphed[0].step_once(sample=step.phase_events is not None)
        coords, target = pipe.current()
        if args.fixed_batch:
            return step.train_step(coords, target, lambda: coords)
        if args.no_prefetch:
            loss = step.train_step(coords, target)
            pipe.produce_next()
        else:
            loss = step.train_step(coords, target, pipe.produce_next, late_work=pipe.produce_late)
        pipe.advance()
        return loss

    def timed_leg(warmup, steps):
        """`warmup` untimed steps, then EXACTLY `steps` timed ones between barrier + synchronize on both
        sides; returns seconds (max over ranks), per-phase ms, host ms per step, the last loss."""
        events.clear()
        sampling[0] = False
        for _ in range(warmup):
            one_step()
        sampling[0], leg_start[0] = True, counter[0]
        if graphed[0] is not None and hasattr(graphed[0], "reserve_samples"):
            graphed[0].reserve_samples(steps // max(1, args.phase_every) + 1)  # no event is created inside the timed region
        # Python's cyclic collector stays out of the timed region: a FULL collection (every object of the process:
        # torch alone brings millions) takes 40-58 ms here, and the allocation counts that trigger it are
        # deterministic -- in the 40-step form it fell on step 24 of every run, 1.3 ms per step instead of 0.51
        # (round 4, tools/stall_probe.py; MRI_STEP_TIMES=1 prints the host time of every queued step)
        gc.collect()
        gc.disable()
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trace = [] if os.environ.get("MRI_STEP_TIMES") else None  # diagnostic: host time of every queued step
        for _ in range(steps):
            t_step = time.perf_counter()
            loss = one_step()
            if trace is not None:
                trace.append(round((time.perf_counter() - t_step) * 1e3, 3))
        if trace is not None:
            print("host ms per queued step:", trace, file=sys.stderr)
        host_ms = (time.perf_counter() - t0) / steps * 1e3  # time to QUEUE a step (host side)
        if graphed[0] is not None:
            graphed[0].finish()
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = parallel.all_reduce_max(time.perf_counter() - t0, dev)
        gc.enable()
        step.phase_events = events
        phases = step.phase_ms()
        n_samples = min((len(v) for v in events.values()), default=0)
        step.phase_events, sampling[0] = None, False
        if hasattr(step, "check_status"):
            step.check_status()  # the overlapped decoder's wait for the lookup must never have timed out
        return dict(elapsed=elapsed, phases=phases, host_ms=host_ms, loss=float(loss),
                    phase_samples=n_samples,
                    ms_per_step=elapsed / steps * 1e3, value=w["batch"] * world * steps / elapsed)

    def replicas_identical():
        """Data parallel: every rank must hold the same parameters, bit for bit."""
        if world == 1:
            return True
        import torch.distributed as dist
        c = step.flat.param.double().sum().reshape(1)
        lo_, hi_ = c.clone(), c.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        return bool((lo_ == hi_).item())

    # ---- the timed legs.  One GPU: the headline leg.  Several: the gradient-exchange variants, back to
    # back in ONE invocation so that a single scaling pass measures them all (the plain all-reduce first:
    # whatever happens in the others, its line exists); `value` is the fastest leg whose replicas stayed
    # identical, and config.parallelism names it.
    if world == 1:
        plan = [("single", None, None)]
    elif args.dp_mode == "auto":
        plan = [("all_reduce_1", "all_reduce", 1), ("all_reduce_4", "all_reduce", 4),
                ("reduce_scatter", "reduce_scatter", 1)]
    else:
        plan = [(args.dp_mode + (f"_{step.grad_buckets}" if args.dp_mode == "all_reduce" else ""),
                 args.dp_mode, step.grad_buckets)]
    legs, partial_line = {}, [None]
    watchdog, current_leg = [None], [None]

    def give_up():
        """A gradient-exchange leg after the first did not finish (it hangs in its first contact with a real
        multi-GPU RCCL): the line of the finished legs is still printed, with an entry naming the leg that hung,
        and EVERY rank exits non-zero -- a collective that hangs is a failed run, to be investigated from that
        record, not a success with a footnote."""
        name, mode, buckets = current_leg[0]
        if rank == 0 and partial_line[0] is not None:
            line = partial_line[0]
            line["aborted"] = (f"leg {name!r} (dp_mode {mode}, grad_buckets {buckets}) did not finish within "
                               f"{args.leg_timeout} s; this line holds the legs that did")
            line.setdefault("collectives", {}).setdefault("legs", {})[name] = dict(
                error=f"timeout after {args.leg_timeout} s", dp_mode=mode, grad_buckets=buckets, identical=False)
            print(json.dumps(line), flush=True)
        os._exit(4)

    def arm_watchdog(leg):  # one budget per leg, so that the record says WHICH leg hung
        import threading
        current_leg[0] = leg
        watchdog[0] = threading.Timer(args.leg_timeout, give_up)
        watchdog[0].daemon = True
        watchdog[0].start()

    def disarm_watchdog():
        if watchdog[0] is not None:
            watchdog[0].cancel()
            watchdog[0] = None

    def finish(best_name, psnr=None, packed=None, cpu=None):
        """The result line from the legs measured so far (rank 0)."""
        best = legs[best_name]
        phases = best["phases"]
        pm = phase_model(w, step, n_params)
        rooflines = {}
        for name in sorted(k for k in phases if k in pm):
            bound, amount = pm[name]
            sec = phases[name] * 1e-3
            if bound == "hbm":
                r = dict(bound="hbm", achieved=amount / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            else:
                r = mfma_roof(step, amount / sec / 1e12)
            r.update(frac=r["achieved"] / r["peak"], traffic=pmc_traffic(args.workload, name, args.records),
                     ms_per_launch=phases[name])
            rooflines[name] = r
        dominant = max(rooflines, key=lambda k: phases[k])
        roof = dict(rooflines[dominant], kernel=dominant)
        mode = best.get("dp_mode")
        result = {
            "metric": "coord-samples/sec (train)", "value": best["value"], "unit": "coord-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": best["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "arithmetic": arithmetic(args.records), "records": args.records,
            "data": data_name + (" -- DIAGNOSTIC --fixed-batch: one batch reused, not a measured line"
                                 if args.fixed_batch else ""),
            "config": {"workload": f"{args.workload}: {'x'.join(map(str, w['shape']))} "
                                   f"{'sample volume' if w.get('sample_volume') else 'analytic phantom'}"
                                   f", {w['model']}, batch {w['batch']} coords per GPU"
                                   + (", even frames trained, odd frames held out"
                                      if w.get("holdout") else ""),
                       "global_batch": w["batch"] * world, "params": n_params,
                       "parallelism": (f"dp{world} z-slab, {mode}"
                                       + (f", {best['grad_buckets']} level group(s)"
                                          if mode == "all_reduce" else ""))
                       if world > 1 else "single GPU"},
            "roofline": roof,
            "rooflines": rooflines,
            "phases_ms": {k: round(v, 4) for k, v in sorted(phases.items())},
            "phases_sampled_every": max(1, args.phase_every),
            "phases_samples": best.get("phase_samples"),  # launches behind each phases_ms / ms_per_launch mean
            "host_queue_ms_per_step": round(best["host_ms"], 4),
            "launch": "one mri_fused_step call per step (phase events recorded inside the call)"
            if graphed[0] is not None else (launch_note[0] or "eager (queued op by op from Python)"),
            "final_loss": best["loss"],
        }
        if world > 1:  # what the first real multi-GPU run needs to explain itself
            result["collectives"] = {
                "backend": dist_info["backend"], "ranks_seen": dist_info["ranks"],
                "rccl_version": dist_info["rccl"],
                "exposed_ms_per_step": round(phases.get("all_reduce", 0.0), 4),
                "note": "exposed = the compute stream's wait for the reductions + per-group Adam "
                        "(phases_ms.all_reduce contains the Adam launches, phases_ms.adam is absent); "
                        "legs[*].groups: bytes of each reduction and the compute stream's wait for it",
                "legs": {k: {"ms_per_step": round(v["ms_per_step"], 4), "value": v["value"],
                             "dp_mode": v["dp_mode"], "grad_buckets": v["grad_buckets"], "launch": v.get("launch"),
                             "replicas_identical": v["identical"],
                             "exposed_ms_per_step": round(v["phases"].get("all_reduce", 0.0), 4),
                             "groups": v["groups"]} for k, v in legs.items()}}
        if step.encoder is not None:  # whole-step HBM fraction as north_star defines it
            e = step.encoder
            per_coord = 4 * e.dim + 4 + 2 * (1 << e.dim) * e.output_dim * 4 + 28.0 * n_params / w["batch"]
            result["step_hbm_frac"] = best["value"] / world * per_coord / (HBM_PEAK_GBS * 1e9)
        if packed is not None:
            result["packed_records"] = packed
        result["device_warmup"] = device_warmup
        result["batch_order"] = "shuffled (the order the epoch's permutation yields; the spatially ordered form of round 3 was removed)"
        if psnr is not None:
            result["psnr"] = psnr
        if cpu is not None:
            result["cpu_baseline"] = cpu
        return result

    def best_leg():
        ok = [k for k, v in legs.items() if v["identical"]]
        if not ok:
            raise SystemExit("no gradient-exchange leg kept the replicas identical: " + json.dumps(
                {k: v["identical"] for k, v in legs.items()}))
        return max(ok, key=lambda k: legs[k]["value"])

    def quality_train():
        """Train on to the fixed step count the PSNR is quoted at; returns (steps, a copy of the parameters there)."""
        while counter[0] < args.psnr_steps:
            one_step()
        if graphed[0] is not None:
            graphed[0].finish()
        return counter[0], [p.detach().clone() for p in model.parameters()]

    def quality_eval(steps, snapshot):
        """PSNR vs ground-truth voxels of the parameters saved at `steps` (outside the timed region)."""
        psnr = None
        if rank == 0:
            if graphed[0] is not None:
                graphed[0].finish()
            with torch.no_grad():
                later = [p.detach().clone() for p in model.parameters()]
                for p, q in zip(model.parameters(), snapshot):
                    p.copy_(q)
                preds = [step.forward(x)[0].clone() for x, _ in datamodules.DeviceLoader(ds, 1 << 20, shuffle=False)]
                psnr = dict(steps=steps, db=trainer.psnr(torch.cat(preds), ds.pixels)
                            if not w["norm_siren"] else
                            trainer.psnr((torch.cat(preds) + 1) / 2, (ds.pixels + 1) / 2))
                if w.get("holdout"):  # frames the network never saw, and the linear-in-t baseline
                    del preds
                    odd = datamodules.MriImage(volume=vol, norm_siren=w["norm_siren"], device=dev,
                                               frames=slice(1, None, 2))
                    held = [step.forward(x)[0].clone() for x, _ in datamodules.DeviceLoader(odd, 1 << 20, shuffle=False)]
                    psnr["heldout_db"] = trainer.psnr(torch.cat(held), odd.pixels)
                    even = ds.pixels.view(ds.shape)
                    n_odd = odd.shape[-1]
                    linear = 0.5 * (even[..., :n_odd] + even[..., 1:n_odd + 1]) \
                        if even.shape[-1] > n_odd else None
                    if linear is not None:
                        psnr["heldout_linear_interp_db"] = trainer.psnr(linear.reshape(-1, 1),
                                                                        odd.pixels)
                for p, q in zip(model.parameters(), later):
                    p.copy_(q)
        return psnr

    # Device state.  After idle an MI355X takes about a second under load to settle its clocks: a 20-step leg that
    # starts cold reads 4 % slower than the same leg a second later (0.541 against 0.518 ms per step, the decoder
    # 0.200 against 0.181; any GPU work warms it, the model's state does not matter, and an idle gap of a few
    # hundred ms -- the first process on a box loading the kernels of the PSNR pass -- cools it again: DESIGN.md 5, EXPERIMENTS.md Part II).
    # Every timed leg of a run must see the same device: on one GPU the quality leg's TRAINING steps (psnr_steps
    # steps of this very workload, ~1 s) run first, the timed legs follow them without a gap, and the PSNR is
    # evaluated at the end from the parameters saved at psnr_steps; with several ranks (where a leg may hang and
    # the line must survive) the first leg's form runs for about a second, untimed, before the legs.
    # --cold-start times the first leg cold.
    warm_first = not args.cold_start and args.psnr_steps > 0 and world == 1
    psnr, quality, device_warmup = None, None, "none (--cold-start)" if args.cold_start else "none"
    capture_graphs()
    if warm_first:
        quality = quality_train()
        device_warmup = f"the quality leg's {quality[0]} training steps of this workload ran before the timed legs"
    elif not args.cold_start and world > 1:
        n_warm = min(4000, max(1, int(1.0 / max(1e-4, timed_leg(2, 8)["ms_per_step"] * 1e-3))))
        timed_leg(0, n_warm)
        device_warmup = f"{n_warm} untimed steps in the first leg's form before the timed legs"
    for i, (name, mode, buckets) in enumerate(plan):
        if mode is not None:
            step.dp_mode, step.grad_buckets, step._bucket_cache = mode, buckets, None
            if graphed[0] is not None:
                graphed[0].finish()
            capture_graphs()  # (the plain all-reduce leg is queued natively, the other forms op by op)
        if i >= 1 and world > 1:
            arm_watchdog((name, mode, buckets))
        try:
            leg = timed_leg(args.warmup, args.steps)
        except Exception as exc:  # a variant may be refused (e.g. shards that do not divide): say so
            disarm_watchdog()
            if i == 0:
                raise
            legs[name] = dict(error=repr(exc), identical=False, value=0.0, ms_per_step=0.0, phases={}, groups=[],
                              dp_mode=mode, grad_buckets=buckets, host_ms=0.0, loss=float("nan"))
            continue
        identical = replicas_identical()  # (a collective too: still under the leg's watchdog)
        disarm_watchdog()
        leg.update(dp_mode=mode, grad_buckets=buckets, identical=identical,
                   launch="native" if graphed[0] is not None else "eager",
                   groups=[dict(bytes=b, wait_ms=round(leg["phases"].get(f"reduce_wait_{j}", 0.0), 4))
                           for j, b in enumerate(getattr(step, "last_group_bytes", []))]
                   if world > 1 else [])
        legs[name] = leg
        if rank == 0 and world > 1:
            partial_line[0] = finish(best_leg()) if any(v["identical"] for v in legs.values()) else None
    best = best_leg()
    if world > 1:  # continue (PSNR steps) in the mode the line reports
        step.dp_mode, step.grad_buckets, step._bucket_cache = legs[best]["dp_mode"], legs[best]["grad_buckets"], None
        if graphed[0] is not None:
            graphed[0].finish()
        capture_graphs()

    if not warm_first and args.psnr_steps > 0:
        quality = quality_train()
        psnr = quality_eval(*quality)
    # the other record format beside the headline (one GPU, grids with two features per level): the same
    # model and loader, the same number of timed steps
    packed = None
    if (world == 1 and step.encoder is not None and step.encoder.n_features_per_level == 2
            and not args.no_records_leg and args.bwd_method != 1):
        other = "packed" if args.records == "f32" else "f32"
        _lib.set_option("bwd_records", RECORDS[other][0])
        capture_graphs()  # (a graph holds the kernels of the format it was captured under)
        leg = timed_leg(min(args.warmup, 20), args.steps)
        _lib.set_option("bwd_records", RECORDS[args.records][0])
        capture_graphs()
        packed = dict(records=other, value=leg["value"], ms_per_step=leg["ms_per_step"],
                      phases_ms={k: round(v, 4) for k, v in sorted(leg["phases"].items())},
                      arithmetic=RECORDS[other][1])
    if warm_first:
        psnr = quality_eval(*quality)
    if world > 1:  # leave the process group together (rank 0 was busy with the PSNR pass)
        import torch.distributed as dist
        parallel.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    cpu = cpu_baseline(w, args.workload) if world == 1 and not args.no_cpu_baseline else None
    print(json.dumps(finish(best, psnr, packed, cpu)), flush=True)


if __name__ == "__main__":
    main()
