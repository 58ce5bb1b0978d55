"""Host-side logic of the package (no GPU): level tables, NIfTI I/O, config, sharding,
model surfaces and state-dict layout, launcher plumbing."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import hashgrid as ohash

from mri_interpolation_amd import config as cfg
from mri_interpolation_amd import encoding, models, nifti, parallel, trainer


@pytest.mark.parametrize("args", [
    (3, 16, 19, 16, 512), (3, 16, 19, 16, 16 * 1.4 ** 15), (4, 16, 19, 16, 16 * 1.4 ** 15),
    (2, 16, 15, 16, 512), (3, 4, 23, (64, 64, 5), (352, 352, 15)),
    (3, 8, 23, (64, 64, 5), (512, 512, 15))])
def test_level_table_matches_oracle(args):
    assert encoding.level_table(*args) == ohash.level_geometry(*args)


def test_encoder_surface_and_state_dict():
    enc = encoding.MultiResHashGrid(3, 16, 2, 19, 16, 512)
    assert (enc.input_dim, enc.output_dim) == (3, 32)
    assert 2 * enc.table.shape[0] == 10435874
    assert enc.levels[3].hashmap_size == 32768 and enc.levels[3].resolution == 32
    assert enc.levels[2].embedding.weight.shape == (15625, 2)
    assert float(enc.table.abs().max()) <= 1e-4
    sd = enc.state_dict()
    assert list(sd) == [f"levels.{i}.embedding.weight" for i in range(16)]
    enc2 = encoding.MultiResHashGrid(3, 16, 2, 19, 16, 512)
    enc2.load_state_dict(sd)
    assert torch.equal(enc2.table, enc.table)
    v2 = encoding.MultiResHashGridV2(3, 4, 1, 23, (64, 64, 5), (352, 352, 15))
    assert v2.sizes == [262144, 274625, 300763, 328509]
    assert torch.equal(v2.levels[1].resolution, torch.tensor([65.0, 65.0, 6.0]))
    with pytest.raises(ValueError):
        encoding.MultiResHashGridV2(4, 4, 1, 23, (64, 64, 5), (352, 352, 15))


def test_model_surfaces_match_reference_keys():
    fx = load_golden("hashmlp_intended")
    net = models.HashMLP(dim_in=3, n_levels=4, n_features_per_level=1, log2_hashmap_size=23,
                         base_resolution=(64, 64, 5), finest_resolution=(352, 352, 15),
                         dim_hidden=64, dim_out=1, n_layers=2)
    ref = [k for k in fx.meta["state_dict_keys"] if not k.startswith("layers.")]
    assert sorted(net.state_dict()) == sorted(ref)
    # a reference checkpoint (with BaseMLP's dead `layers.*` entries) loads
    sd = dict(net.state_dict())
    sd["layers.0.weight"], sd["layers.0.bias"] = torch.zeros(128, 2), torch.zeros(128)
    net.load_state_dict(sd)
    siren = models.SirenNet(dim_in=3, dim_hidden=256, dim_out=1, n_layers=5)
    assert sum(p.numel() for p in siren.parameters()) == 264449
    assert list(siren.state_dict())[:2] == ["layers.0.weight", "layers.0.bias"]
    assert "last_layer.weight" in siren.state_dict()
    w = siren.layers[0].weight
    assert float(w.abs().max()) <= 1 / 3 and float(siren.layers[1].weight.abs().max()) <= \
        np.sqrt(6 / 256) / 30
    mlp = models.BaseMLP(dim_in=32, dim_out=1, dim_hidden=64, n_layers=3)
    assert sorted(mlp.state_dict()) == sorted(f"layers.{2 * i}.{p}" for i in range(3)
                                              for p in ("weight", "bias"))
    assert trainer.fusable_layers(siren) is not None and trainer.fusable_layers(net) is None
    tiny = models.HashMLP(3, 16, 2, 19, 16, 512, dim_hidden=64, n_layers=3,
                          activation=torch.nn.ReLU, batch_norm=False, final_activation=False)
    enc, layers = trainer.fusable_layers(tiny)
    assert enc is tiny.encoder and [l.weight.shape for l in layers] == [(64, 32), (64, 64), (1, 64)]
    assert sum(p.numel() for p in tiny.parameters()) == 10442211  # SURVEY 8(d) cfg 2


def test_cpu_forward_is_refused():
    net = models.SirenNet(dim_in=2, dim_hidden=16, dim_out=1, n_layers=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(4, 2))


def test_nifti_round_trip_and_sample_slice(tmp_path):
    vol = (np.arange(6 * 5 * 4 * 3, dtype=np.float32).reshape(6, 5, 4, 3) - 17.0) * 0.5
    for name in ("a.nii.gz", "b.nii"):
        path = str(tmp_path / name)
        nifti.save(vol, path)
        assert nifti.read_header(path)["shape"] == (6, 5, 4, 3)
        np.testing.assert_array_equal(nifti.load(path), vol)
    # int16 + scl_slope, Fortran order: the sample's storage format
    fx = load_golden("sample_slice_z3_t7")
    raw = fx["raw_int16"]
    path = str(tmp_path / "s.nii.gz")
    nifti.save(raw, path)
    np.testing.assert_array_equal(nifti.load(path), raw.astype(np.float32))
    sample = "/root/reference/sample_ankle_dyn_mri.nii.gz"
    if os.path.exists(sample):  # only in the build container
        full = nifti.load(sample)
        assert full.shape == (352, 352, 6, 15)
        want = (raw.astype(np.float64) * fx.meta["scl_slope"]).astype(np.float32)
        np.testing.assert_array_equal(full[:, :, 3, 7], want)


def test_config_defaults_and_json_mapping(tmp_path):
    c = cfg.HashConfig().resolve((352, 352, 15))
    assert (c.dim_in, c.n_levels, c.n_features_per_level, c.log2_hashmap_size) == (3, 4, 1, 23)
    assert c.base_resolution == (64, 64, 5) and c.lr == 5e-3 and c.batch_size == 10000
    b = cfg.BaseConfig()
    assert (b.dim_hidden, b.n_layers, b.batch_size, b.lr, b.w0) == (128, 6, 4096, 1e-4, 30.0)
    enc = cfg.encoder_from_json(cfg.load_json(os.path.join(ROOT, "config", "hash_config.json")), 3)
    res, sizes = encoding.level_table(3, enc["n_levels"], enc["log2_hashmap_size"],
                                      enc["base_resolution"], enc["finest_resolution"])
    assert [r[0] for r in res][-3:] == [1269, 1777, 2489] and 2 * sum(sizes) == 12236382
    cfg.apply_overrides(c, dict(batch_size=77, epochs=None))
    assert c.batch_size == 77 and c.epochs == 1
    c.export_to_txt(str(tmp_path))
    assert "batch_size : 77" in open(tmp_path / "config.txt").read()
    assert cfg.parse_slice(":,:,3,7") == (slice(None), slice(None), 3, 7)


def test_sharding_ranges():
    spans = [parallel.voxel_range((256, 256, 256), r, 8) for r in range(8)]
    assert spans[0] == (0, 32 * 65536) and spans[-1][1] == 256 ** 3
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # the sample has 6 z-slices... and 352 x-slices: slowest axis (x) is sharded
    spans = [parallel.voxel_range((352, 352, 6, 15), r, 8) for r in range(8)]
    assert spans[3] == (3 * 44 * 352 * 90, 4 * 44 * 352 * 90)
    # fewer slices than ranks: contiguous flat split
    spans = [parallel.voxel_range((6, 100), r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 600 and all(b - a == 75 for a, b in spans)
    assert [parallel.slab_range(6, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 6)]
    with pytest.raises(ValueError):
        parallel.slab_range(6, 0, 8)


def test_launcher_cli_and_model_factory():
    import launcher
    args = launcher.parse_args(["--batch_size", "4096", "--epochs", "2", "--model_class",
                                "SirenNet", "--slice", ":,:,3,7"])
    assert (args.batch_size, args.epochs, args.model_class, args.slice_spec) == \
        (4096, 2, "SirenNet", ":,:,3,7")
    c = cfg.apply_overrides(cfg.BaseConfig(), dict(model_class="SirenNet")).resolve((352, 352))
    net = launcher.build_model(c, models)  # SirenNet takes no hash kwargs
    assert isinstance(net, models.SirenNet) and net.layers[0].weight.shape == (128, 2)
    h = cfg.HashConfig().resolve((352, 352, 15))
    assert isinstance(launcher.build_model(h, models), models.HashMLP)


def test_interp_baseline_is_linear_in_time():
    import interp
    t = np.arange(15, dtype=np.float64)
    data = np.broadcast_to(3.0 * t + 1.0, (4, 5, 15)).copy()
    out = interp.interpolate_even_frames(data)
    np.testing.assert_allclose(out, data, rtol=0, atol=1e-12)  # exact on a linear ramp
    data2 = np.random.default_rng(0).random((3, 3, 8))
    out2 = interp.interpolate_even_frames(data2)
    np.testing.assert_array_equal(out2[..., ::2], data2[..., ::2])
    np.testing.assert_allclose(out2[..., 1], 0.5 * (data2[..., 0] + data2[..., 2]))
    np.testing.assert_array_equal(out2[..., 7], data2[..., 6])  # beyond the last even frame


def test_phantom_4d_is_a_moving_3d_phantom():
    """phantom_volume with an (x, y, z, t) shape: normalised to [0, 1], frames differ, and a
    3-D shape still gives the static phantom of SURVEY.md 8(d)."""
    import torch
    from mri_interpolation_amd.datamodules import phantom_volume
    v4 = phantom_volume((12, 10, 6, 5), device="cpu")
    assert v4.shape == (12, 10, 6, 5) and v4.dtype == torch.float32
    assert float(v4.min()) == 0.0 and float(v4.max()) == 1.0
    assert not torch.allclose(v4[..., 0], v4[..., 2])
    v3 = phantom_volume((12, 10, 6), device="cpu")
    assert v3.shape == (12, 10, 6) and float(v3.min()) == 0.0 and float(v3.max()) == 1.0



def test_resolution_beyond_int64_raises_like_the_reference():
    """base_resolution 2 makes the growth factor finest / base per level: level 11
    of this grid has a resolution of 1.9e23.  The reference multiplies the coordinates by that
    Python int and torch raises OverflowError at the first forward; so does the drop-in (before
    anything touches the GPU).  Per-axis (V2) grids hold float32 resolutions and never raise."""
    enc = encoding.MultiResHashGrid(2, 12, 2, 8, 2, 245)
    assert enc.resolutions[-1][0] > 2 ** 63 and list(enc.sizes)[-1] == 256
    with pytest.raises(OverflowError, match="int too big to convert"):
        enc(torch.rand(4, 2))
    assert not encoding.MultiResHashGrid(2, 4, 2, 8, 2, 245)._too_fine
    assert not encoding.MultiResHashGridV2(2, 12, 2, 8, (2, 2), (245, 245))._too_fine


# ----------------------------------------------------------------- data-parallel loaders (host)
class _FakeDataset:
    """Shape-only stand-in: the loader arithmetic under test never touches the GPU."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.dim_in = len(shape)
        self.device = "cpu"

    def __len__(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


@pytest.mark.parametrize("shape,world,batch", [((100, 100, 100), 8, 10000),
                                               ((150, 256, 256), 8, 65536),
                                               ((6, 40, 40), 8, 500),      # flat-range fallback
                                               ((352, 352, 6, 8), 8, 1 << 18)])
def test_sharded_loaders_run_equal_steps(shape, world, batch):
    """Slabs differ in size whenever the slow axis does not divide by the world size (the
    first two cases gave 13 vs 12 and 19 vs 18 batches per epoch before): every rank must still
    run the same number of equal-sized steps, or some ranks leave the all-reduce early."""
    from mri_interpolation_amd import datamodules, parallel
    ds = _FakeDataset(shape)
    loaders = [datamodules.sharded_loader(ds, batch, r, world) for r in range(world)]
    steps = {len(l) for l in loaders}
    assert len(steps) == 1
    n_steps = steps.pop()
    spans = [parallel.voxel_range(shape, r, world) for r in range(world)]
    assert n_steps == -(-max(hi - lo for lo, hi in spans) // batch)
    for l in loaders:
        assert [l.span(b)[1] for b in range(n_steps)] == [batch] * n_steps
        assert l.seed != loaders[(loaders.index(l) + 1) % world].seed
    # slabs tile the volume
    assert spans[0][0] == 0 and spans[-1][1] == len(ds)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # one process keeps the reference's loader: last batch of an epoch may be short
    single = datamodules.sharded_loader(ds, batch, 0, 1)
    assert single.steps is None and len(single) == -(-len(ds) // batch)


def test_accumulate_schedule():
    from mri_interpolation_amd import trainer
    assert trainer._accumulate_schedule(None) == {0: 1}
    assert trainer._accumulate_schedule(4) == {0: 4}
    sched = trainer._accumulate_schedule({200: 2, 5: 3})   # reference config/base.py:27 form
    assert list(sched.items()) == [(0, 1), (5, 3), (200, 2)]
    assert [trainer._accumulate_at(sched, e) for e in (0, 4, 5, 199, 200, 999)] == [1, 1, 3, 3, 2, 2]
    with pytest.raises(ValueError):
        trainer._accumulate_schedule(0)
    with pytest.raises(TypeError):
        trainer._accumulate_schedule("often")


def test_trainer_rejects_unknown_arguments_and_missing_group(monkeypatch):
    from mri_interpolation_amd import trainer
    with pytest.raises(TypeError):
        trainer.Trainer(gradient_clip_val=0.5)   # silently swallowed before
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(RuntimeError, match="no process group"):
        trainer.Trainer()                        # would pre-divide gradients by 4 and never reduce
    assert trainer.Trainer(distributed=False).world == 1


def test_entry_points_set_the_ipc_mode_before_torch_loads():
    """HSA_ENABLE_IPC_MODE_LEGACY is read when HIP initialises: bench.py and launcher.py (the
    programs torchrun starts, one per GPU) must default it before anything imports torch."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("bench.py", "launcher.py"):
        tree = ast.parse(open(os.path.join(root, name)).read())
        seen_env = False
        for node in tree.body:  # module level, in order
            src = ast.unparse(node)
            if "HSA_ENABLE_IPC_MODE_LEGACY" in src:
                seen_env = True
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
                assert seen_env or not any(m.split(".")[0] in ("torch", "mri_interpolation_amd")
                                           for m in mods), f"{name}: torch imported before the default"
        assert seen_env, name


def test_ipc_mode_is_defaulted_for_multi_process_launches_only(monkeypatch):
    """A single process never touches HSA_ENABLE_IPC_MODE_LEGACY; with WORLD_SIZE > 1 it is a default
    that an explicit setting wins over (parallel.ipc_default; bench.py / launcher.py do the same at
    module level, before torch loads)."""
    from mri_interpolation_amd import parallel
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    parallel.ipc_default()
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ
    parallel.ipc_default(1)
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ
    monkeypatch.setenv("WORLD_SIZE", "8")
    parallel.ipc_default()
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    parallel.ipc_default(8)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("bench.py", "launcher.py"):
        src = open(os.path.join(root, name)).read()
        guard = src.index('os.environ.get("WORLD_SIZE", "1")) > 1')
        assert guard < src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY"'), name


@pytest.mark.parametrize("frames", [15, 14])
def test_interp_baseline_equals_scipy_linear_interpolation(frames):
    """interp.py's counterpart (reference interp.py:35-50: even frames -> itk.LinearInterpolateImageFunction,
    evaluated at (t / 2, y, x) for every voxel) against an INDEPENDENT implementation of the same linear
    interpolation, `scipy.ndimage.map_coordinates(order=1)` (ITK itself is not installed), on
    the reference's own slice [:, :, 3, :] of the sample volume, for an odd and an even number of frames (an
    even one puts the last odd frame half a step beyond the last even frame: ITK clamps the upper
    neighbour to the buffer's end, `mode="nearest"` does the same)."""
    import interp
    from scipy import ndimage
    fx = load_golden("sample_volume")
    vol = fx["raw_int16"][:, :, 3, :frames].astype(np.float32) * np.float32(fx.meta["scl_slope"])
    data = vol / vol.max()                                        # interp.py:26-27
    values = data[..., ::2].astype(np.float64)                    # interp.py:35
    ix, iy, it = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in data.shape], indexing="ij")
    want = ndimage.map_coordinates(values, [ix, iy, it / 2.0], order=1, mode="nearest")
    got = interp.interpolate_even_frames(data)
    assert got.shape == data.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12
    assert np.array_equal(got[..., ::2], values)                  # the trained frames are reproduced exactly
    # and the PSNR figure bench.py reports beside the network's (held-out odd frames)
    odd = slice(1, None, 2)
    mse = np.mean((got[..., odd] - data[..., odd].astype(np.float64)) ** 2)
    assert abs(interp.psnr(got[..., odd], data[..., odd]) - 10.0 * np.log10(1.0 / mse)) <= 1e-9
