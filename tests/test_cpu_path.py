"""The launcher's GPU-less mode (BASELINE config 1: SirenNet on one 2-D slice of the sample volume, PyTorch CPU;
reference launcher.py:157 falls back to the CPU when no GPU is visible).  Plumbing: it must run end to end without
the MI355X library and leave the reference's artefacts behind."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def slice_nifti(tmp_path):
    """The 352 x 352 slice [:, :, 3, 7] of the reference's sample volume (tests/golden/sample_volume.npz), as NIfTI."""
    from mri_interpolation_amd import nifti
    import json
    z = np.load(os.path.join(ROOT, "tests", "golden", "sample_volume.npz"))
    vol = z["raw_int16"].astype(np.float32) * float(json.loads(str(z["meta"])).get("scl_slope", 1.0))
    path = str(tmp_path / "slice.nii.gz")
    nifti.save(np.ascontiguousarray(vol[:, :, 3, 7]), path)
    return path


def test_cpu_siren_equals_the_oracle_on_the_same_batch():
    """cpu_path.siren_forward is the reference's SirenNet.forward: equal to the oracle's restatement on the
    product module's own parameters (and the coordinate grid / normalisation to the oracle's data restatement)."""
    from mri_interpolation_amd import cpu_path, models
    from oracle import data as odata
    from oracle import mlp as omlp
    torch.manual_seed(3)
    net = models.SirenNet(dim_in=2, dim_hidden=32, dim_out=1, n_layers=3)
    x = torch.rand(50, 2) * 2 - 1
    params = [(l.weight.detach(), l.bias.detach()) for l in list(net.layers) + [net.last_layer]]
    assert torch.equal(cpu_path.siren_forward(net, x), omlp.siren_forward(x, params))
    vol = np.arange(6 * 5, dtype=np.float32).reshape(6, 5) ** 1.5
    c, p = odata.dataset(vol, norm_siren=True)
    assert torch.equal(cpu_path.grid_coords(vol.shape), c)
    assert torch.equal(cpu_path.normalised_pixels(vol), p.reshape(-1, 1))


def test_launcher_runs_config_1_without_a_gpu(slice_nifti, tmp_path, monkeypatch):
    """`launcher.py --accelerator cpu --model_class SirenNet` on the 2-D slice: trains (the loss falls), predicts,
    writes pred.nii.gz / pred.npy / config.txt and a checkpoint the GPU path's loader reads -- and never loads the
    HIP library."""
    import launcher
    from mri_interpolation_amd import _lib, checkpoint, models, nifti
    monkeypatch.setattr(_lib, "load", lambda: (_ for _ in ()).throw(AssertionError("the CPU mode must not load the library")))
    out = str(tmp_path / "run")
    launcher.main(["--model_class", "SirenNet", "--image_path", slice_nifti, "--batch_size", "8192", "--epochs", "2",
                   "--dim_hidden", "64", "--n_layers", "3", "--lr", "1e-3", "--accelerator", "cpu", "--out_dir", out,
                   "--log_every", "0"])
    assert os.path.exists(os.path.join(out, "pred.nii.gz")) and os.path.exists(os.path.join(out, "pred.npy"))
    pred = nifti.load(os.path.join(out, "pred.nii.gz"))
    assert pred.shape == (352, 352) and np.isfinite(pred).all()
    cfg = dict(l.rstrip("\n").split(" : ", 1) for l in open(os.path.join(out, "config.txt")) if l.count(" : ") >= 1
               and not l.startswith(" "))
    assert cfg["accelerator"] == "cpu" and cfg["model_class"] == "SirenNet" and float(cfg["psnr_db"]) > 8.0
    ckpts = os.listdir(os.path.join(out, "checkpoints"))
    assert len(ckpts) == 1 and ckpts[0].startswith("epoch=1-step=")
    net = models.SirenNet(dim_in=2, dim_hidden=64, dim_out=1, n_layers=3)
    checkpoint.load(os.path.join(out, "checkpoints", ckpts[0]), net)
    with pytest.raises(SystemExit, match="SirenNet only"):
        launcher.main(["--model_class", "HashMLP", "--image_path", slice_nifti, "--accelerator", "cpu", "--out_dir", out])
