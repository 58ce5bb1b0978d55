#!/usr/bin/env python3
"""Launcher: train an implicit representation of one MRI volume, predict it back, interpolate
it on denser grids, save NIfTI outputs.

Keeps the reference entry point `launcher.py` (same CLI flags, reference launcher.py:36-59, and
the same artefacts: `pred.nii.gz`, `interpolation{shape}.nii.gz`, `config.txt`,
launcher.py:179-224) on top of the MI355X hot path.  Known defects of the reference script are
resolved as listed in SURVEY.md section 0 (Q5 keyword filtering per model class, Q6 output
directory / undefined names / removed Lightning arguments).

    python launcher.py --model_class HashMLP --epochs 1 --batch_size 262144
    torchrun --nproc-per-node 8 launcher.py ...        # z-slab data parallel, RCCL
"""
import argparse
import inspect
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL's dmabuf IPC between the ranks of a node is read when HIP initialises: set before torch loads,
# and only for a multi-process launch (mri_interpolation_amd/parallel.py: ipc_default)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    # reference flags (launcher.py:36-59)
    p.add_argument("--batch_size", help="batch size", type=int)
    p.add_argument("--epochs", help="Number of epochs", type=int)
    p.add_argument("--accumulate_grad_batches", type=int,
                   help="number of batches accumulated per gradient descent step")
    p.add_argument("--n_sample", help="number of points for psf in x, y, z", type=int)
    p.add_argument("--model_class", help="Model class selection", type=str)
    p.add_argument("--enco_config_path", help="path for the encoding config json", type=str)
    # additions
    p.add_argument("--image_path", type=str)
    p.add_argument("--slice", dest="slice_spec", type=str,
                   help="train on a sub-volume, e.g. ':,:,3,7' or ':,:,3,:'")
    p.add_argument("--synthetic", type=str, help="use the analytic phantom, e.g. 128,128,128")
    p.add_argument("--lr", type=float)
    p.add_argument("--dim_hidden", type=int)
    p.add_argument("--n_layers", type=int)
    p.add_argument("--base_resolution", type=str,
                   help="hash grid: one integer, or one per axis, e.g. 16,16,5,7 (per-axis values "
                        "select MultiResHashGridV2, as in reference models.py:691-708)")
    p.add_argument("--finest_resolution", type=str, help="as --base_resolution (floats allowed)")
    p.add_argument("--tiny_mlp", action="store_true",
                   help="HashMLP with the fused ReLU tiny-MLP decoder of hash_config.json")
    p.add_argument("--holdout_odd_frames", action="store_true",
                   help="train on the even frames of the last axis, report PSNR on the odd ones "
                        "(BASELINE config 5 protocol)")
    p.add_argument("--checkpoint_path", type=str,
                   help="resume the parameters from a checkpoint (what the reference does through "
                        "model_cls.load_from_checkpoint when config.checkpoint_path is set)")
    p.add_argument("--resume_optimizer", action="store_true",
                   help="with --checkpoint_path: also restore Adam's moments and step count (Lightning's "
                        "fit(ckpt_path=)); the default is the reference's resume, a fresh Adam at --lr")
    p.add_argument("--restore_lr", action="store_true",
                   help="with --resume_optimizer: take the checkpoint's learning rate instead of --lr")
    p.add_argument("--unsafe_checkpoint", action="store_true",
                   help="read --checkpoint_path with the unrestricted unpickler (Lightning files holding "
                        "callback objects); executes code from the file")
    p.add_argument("--accelerator", choices=("auto", "gpu", "cpu"), default="auto",
                   help="auto: the MI355X path if a GPU is visible, else the PyTorch-CPU plumbing mode (SirenNet only; "
                        "reference launcher.py:157: accelerator='gpu' if torch.cuda.is_available() else 'cpu')")
    p.add_argument("--out_dir", type=str, default=None)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--log_every", type=int, default=50)
    return p.parse_args(argv)


def build_model(config, models):
    """Instantiate `config.model_class`, passing only the keywords it accepts (Q5)."""
    import torch
    cls = getattr(models, config.model_class)
    everything = dict(
        dim_in=config.dim_in, dim_hidden=config.dim_hidden, dim_out=config.dim_out,
        n_layers=config.n_layers, w0=config.w0, w0_initial=config.w0_initial,
        use_bias=config.use_bias, final_activation=config.final_activation, lr=config.lr)
    if config.model_class == "HashMLP":
        act = getattr(torch.nn, getattr(config, "activation", "GELU"))
        everything.update(
            n_levels=config.n_levels, n_features_per_level=config.n_features_per_level,
            log2_hashmap_size=config.log2_hashmap_size, base_resolution=config.base_resolution,
            finest_resolution=config.finest_resolution, dropout=config.dropout, activation=act,
            batch_norm=config.batch_norm)
        everything.pop("final_activation")
        everything["final_activation"] = config.final_activation_on
    params = inspect.signature(cls.__init__).parameters
    takes_kwargs = any(p.kind == p.VAR_KEYWORD for p in params.values())
    kwargs = {k: v for k, v in everything.items() if takes_kwargs or k in params}
    return cls(**kwargs)


def main(argv=None):
    args = parse_args(argv)
    import numpy as np
    import torch
    from mri_interpolation_amd import _lib, config as cfg, datamodules, models, nifti, parallel
    from mri_interpolation_amd.trainer import Trainer, psnr

    # reference launcher.py:157: the GPU when there is one, else the CPU.  The CPU mode is plumbing (BASELINE config 1):
    # plain PyTorch, SirenNet only (mri_interpolation_amd/cpu_path.py); the MI355X path itself has no CPU form and a
    # GPU run without its library still fails loudly
    use_cpu = args.accelerator == "cpu" or (args.accelerator == "auto" and not torch.cuda.is_available())
    torch.manual_seed(1337)  # reference launcher.py:30
    if use_cpu:
        if (args.model_class or "HashMLP") != "SirenNet":
            raise SystemExit(f"--accelerator cpu trains SirenNet only (BASELINE config 1); {args.model_class or 'HashMLP'} "
                             "runs on the MI355X path (--accelerator gpu)")
        rank, world, local = 0, 1, 0
    else:
        _lib.load()
        rank, world, local = parallel.init()
        torch.cuda.set_device(local)

    wants_hash = (args.model_class or "HashMLP") == "HashMLP"
    config = cfg.HashConfig() if wants_hash else cfg.BaseConfig()
    enco_path = args.enco_config_path or os.path.join(ROOT, "config", "hash_config.json")
    if os.path.exists(enco_path):
        config.enco_config = cfg.load_json(enco_path)  # reference launcher.py:73-74
    overrides = {k: v for k, v in vars(args).items()
                 if k not in ("synthetic", "tiny_mlp", "out_dir", "max_steps", "log_every",
                              "resume_optimizer", "restore_lr", "unsafe_checkpoint", "accelerator",
                              "enco_config_path", "holdout_odd_frames", "base_resolution",
                              "finest_resolution")}
    cfg.apply_overrides(config, overrides)

    # ---- data ---------------------------------------------------------------------------
    if args.synthetic:
        shape = tuple(int(s) for s in args.synthetic.split(","))
        volume = datamodules.phantom_volume(shape).cpu().numpy()
    else:
        volume = nifti.load(config.image_path)
        if config.slice_spec:
            volume = volume[cfg.parse_slice(config.slice_spec)]
    config.resolve(volume.shape)
    if wants_hash and args.tiny_mlp:
        enc = cfg.encoder_from_json(config.enco_config or {}, config.dim_in)
        net = (config.enco_config or {}).get("network", {})
        config.n_levels, config.n_features_per_level = enc["n_levels"], enc["n_features_per_level"]
        config.log2_hashmap_size = enc["log2_hashmap_size"]
        config.base_resolution, config.finest_resolution = enc["base_resolution"], enc["finest_resolution"]
        config.dim_hidden = args.dim_hidden or int(net.get("n_neurons", 128))
        config.n_layers = args.n_layers or int(net.get("n_hidden_layers", 2)) + 1
        config.activation, config.batch_norm, config.final_activation_on = "ReLU", False, False
    for name in ("base_resolution", "finest_resolution"):  # explicit grids win over the JSON's
        if wants_hash and getattr(args, name):
            setattr(config, name, cfg.parse_resolution(getattr(args, name)))
    if wants_hash and not isinstance(config.base_resolution, int) \
            and len(config.base_resolution) != config.dim_in:
        raise SystemExit(f"base_resolution {config.base_resolution} does not match the "
                         f"{config.dim_in}-D volume (SURVEY.md Q7): pass --slice, --tiny_mlp or "
                         "--base_resolution / --finest_resolution with one value per axis")
    config.norm_siren = config.model_class in ("SirenNet", "ModulatedSirenNet")
    if use_cpu:
        return main_cpu(args, config, volume)

    model = build_model(config, models).cuda()
    if config.checkpoint_path:  # reference launcher.py:97-117 (model_cls.load_from_checkpoint)
        from mri_interpolation_amd import checkpoint
        # The reference restores the WEIGHTS and then fits with a fresh Adam at config.lr (no ckpt_path in
        # trainer.fit, launcher.py:165): that is the default here too.  --resume_optimizer restores the
        # moments and the step count as Lightning's fit(ckpt_path=) would (Trainer.fit reuses model.optimizer).
        model.optimizer = model.configure_optimizers()
        checkpoint.load(config.checkpoint_path, model, model.optimizer, map_location="cuda",
                        resume_optimizer=args.resume_optimizer, restore_lr=args.restore_lr,
                        allow_pickle=args.unsafe_checkpoint)
    datamodule = datamodules.MriDataModule(config=config, volume=volume,
                                           norm_siren=config.norm_siren)
    datamodule.prepare_data()
    datamodule.setup()
    train_loader = datamodule.train_dataloader(rank, world)
    test_loader = datamodule.test_dataloader()
    if args.holdout_odd_frames:  # train on even frames, coordinates from the full time grid
        even = datamodules.MriImage(volume=volume, norm_siren=config.norm_siren,
                                    frames=slice(0, None, 2))
        train_loader = datamodules.sharded_loader(even, config.batch_size, rank, world,
                                                  seed=config.seed)

    # ---- training -------------------------------------------------------------------------
    trainer = Trainer(max_epochs=config.epochs, max_steps=args.max_steps, precision=32,
                      log_every=args.log_every,
                      accumulate_grad_batches=config.accumulate_grad_batches)
    t0 = time.time()
    trainer.fit(model, train_loader)
    train_seconds = time.time() - t0
    if world > 1:  # every rank leaves the group together; rank 0 alone writes the artefacts
        if getattr(model, "optimizer", None) is not None:
            parallel.gather_optimizer_state(model.optimizer, rank, world)  # reduce_scatter: moments made whole
        parallel.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return

    # ---- prediction and outputs (reference launcher.py:173-224) ---------------------------
    out_dir = args.out_dir
    if out_dir is None:
        base = os.path.join(ROOT, "lightning_logs")
        version = 0
        while os.path.exists(os.path.join(base, f"version_{version}")):
            version += 1
        out_dir = os.path.join(base, f"version_{version}")
        config.log = str(version)
    os.makedirs(out_dir, exist_ok=True)

    # Lightning's default checkpoint, lightning_logs/version_N/checkpoints/epoch=E-step=S.ckpt, in the
    # layout Lightning writes (checkpoint.py): the reference's load_from_checkpoint reads it
    from mri_interpolation_amd import checkpoint
    os.makedirs(os.path.join(out_dir, "checkpoints"), exist_ok=True)
    if rank == 0:  # (a sharded optimiser state was gathered before the ranks parted, below)
        checkpoint.save(os.path.join(out_dir, "checkpoints",
                                     f"epoch={config.epochs - 1}-step={trainer.global_step}.ckpt"),
                        model, getattr(model, "optimizer", None), epoch=config.epochs - 1,
                        global_step=trainer.global_step)

    pred = torch.concat(trainer.predict(model, test_loader))
    truth = datamodule.dataset.pixels
    quality = psnr((pred + 1) / 2, (truth + 1) / 2) if config.norm_siren else psnr(pred, truth)
    im = np.array(pred.reshape(config.image_shape).detach().cpu().numpy(), dtype=np.float32)
    if im.ndim == 2:
        np.save(os.path.join(out_dir, "pred.npy"), im)
    nifti.save(im, os.path.join(out_dir, "pred.nii.gz"))

    for shape in config.interp_shapes:
        if len(shape) != config.dim_in:
            print(f"skip interpolation shape {shape}: volume is {config.dim_in}-D")
            continue
        loader = datamodule.upsampling(shape, config.batch_size, norm_siren=config.norm_siren)
        interp = torch.concat(trainer.predict(model, loader))
        interp_im = np.array(interp.reshape(shape).detach().cpu().numpy(), dtype=np.float32)
        nifti.save(interp_im, os.path.join(out_dir, f"interpolation{tuple(shape)}.nii.gz"))

    if args.holdout_odd_frames:
        odd = datamodules.MriImage(volume=volume, norm_siren=config.norm_siren,
                                   frames=slice(1, None, 2))
        held = torch.concat(trainer.predict(
            model, datamodules.DeviceLoader(odd, config.batch_size, shuffle=False)))
        config.psnr_heldout_db = (psnr((held + 1) / 2, (odd.pixels + 1) / 2)
                                  if config.norm_siren else psnr(held, odd.pixels))
        print(f"PSNR on the {odd.shape[-1]} held-out odd frames: {config.psnr_heldout_db:.2f} dB")
    config.train_seconds = train_seconds
    config.psnr_db = quality
    config.coords_per_second = trainer.throughput[-1] if trainer.throughput else None
    config.export_to_txt(out_dir)
    print(f"trained {trainer.global_step} steps in {train_seconds:.2f} s "
          f"({config.coords_per_second:.3e} coord-samples/s), PSNR {quality:.2f} dB -> {out_dir}")


def main_cpu(args, config, volume):
    """Train, predict and write the artefacts without a GPU: plain PyTorch on the CPU (cpu_path.py), SirenNet."""
    import numpy as np
    import torch
    from mri_interpolation_amd import checkpoint, cpu_path, models, nifti
    model = build_model(config, models)
    if config.checkpoint_path:
        checkpoint.load(config.checkpoint_path, model, allow_pickle=args.unsafe_checkpoint)
    coords = cpu_path.grid_coords(volume.shape, norm_siren=True)
    pixels = cpu_path.normalised_pixels(volume, norm_siren=True)
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    losses, steps = cpu_path.fit(model, coords, pixels, config.batch_size, config.epochs, seed=config.seed,
                                 max_steps=args.max_steps, log_every=args.log_every)
    train_seconds = time.time() - t0
    out_dir = args.out_dir
    if out_dir is None:
        base, version = os.path.join(ROOT, "lightning_logs"), 0
        while os.path.exists(os.path.join(base, f"version_{version}")):
            version += 1
        out_dir = os.path.join(base, f"version_{version}")
        config.log = str(version)
    os.makedirs(os.path.join(out_dir, "checkpoints"), exist_ok=True)
    checkpoint.save(os.path.join(out_dir, "checkpoints", f"epoch={config.epochs - 1}-step={steps}.ckpt"), model,
                    epoch=config.epochs - 1, global_step=steps)
    pred = cpu_path.predict(model, coords, config.batch_size)
    quality = cpu_path.psnr((pred + 1) / 2, (pixels + 1) / 2)
    im = np.array(pred.reshape(config.image_shape).numpy(), dtype=np.float32)
    if im.ndim == 2:
        np.save(os.path.join(out_dir, "pred.npy"), im)
    nifti.save(im, os.path.join(out_dir, "pred.nii.gz"))
    for shape in config.interp_shapes:
        if len(shape) != config.dim_in:
            print(f"skip interpolation shape {shape}: volume is {config.dim_in}-D")
            continue
        interp = cpu_path.predict(model, cpu_path.grid_coords(shape, norm_siren=True), config.batch_size)
        nifti.save(np.array(interp.reshape(shape).numpy(), dtype=np.float32),
                   os.path.join(out_dir, f"interpolation{tuple(shape)}.nii.gz"))
    config.train_seconds, config.psnr_db, config.accelerator = train_seconds, quality, "cpu"
    config.coords_per_second = steps * config.batch_size / max(train_seconds, 1e-9)
    config.final_loss = losses[-1] if losses else None
    config.export_to_txt(out_dir)
    print(f"trained {steps} steps on {torch.get_num_threads()} CPU threads in {train_seconds:.2f} s "
          f"({config.coords_per_second:.3e} coord-samples/s), PSNR {quality:.2f} dB -> {out_dir}")


if __name__ == "__main__":
    main()
