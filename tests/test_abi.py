"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/mri_inr.h declares; argument validation works before any HIP call.  CPU only."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mri_inr.h")


@pytest.fixture(scope="module")
def lib():
    from mri_interpolation_amd import _lib
    from mri_interpolation_amd.build import build
    build()
    return _lib


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mri_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 14, names
    handle = lib.load()
    for name in names:
        assert hasattr(handle, name), f"{name} declared in mri_inr.h but not exported"
    bound = (set(lib.SIGNATURES) | set(lib.STRING_GETTERS) | set(lib.INT64_GETTERS)
             | set(lib.INT_GETTERS))
    assert bound == set(names), (bound ^ set(names))


def test_version_and_struct_layout(lib):
    assert lib.version().startswith("mri_inr") and "gfx950" in lib.version()
    # struct mri_grid_desc: 4 int32 + float[32][8] + uint32[32] + uint64[32]
    assert C.sizeof(lib.GridDesc) == 16 + 32 * 8 * 4 + 32 * 4 + 32 * 8


def test_argument_validation_without_gpu(lib):
    from mri_interpolation_amd import ops
    desc = ops.make_grid_desc(3, [[16, 16, 16]], [4096], 2)
    h = lib.load()
    # empty batches are a no-op and touch no device
    assert h.mri_hashgrid_forward(C.byref(desc), None, 0, None, None, 2, 2, 1, None) == 0
    assert h.mri_linear_forward(None, 1, 1, None, None, 0, 4, 4, 0, 1.0, None, 4, None, 0, None) == 0
    bad = ops.make_grid_desc(3, [[16, 16, 16]], [4096], 2)
    bad.n_features = 3
    assert h.mri_hashgrid_forward(C.byref(bad), None, 0, None, None, 2, 2, 1, None) == -1
    assert "n_features" in h.mri_last_error().decode()
    assert h.mri_linear_forward(None, 1, 1, None, None, 8, 0, 4, 0, 1.0, None, 4, None, 0, None) == -1
    assert h.mri_adam_step(None, None, None, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None) == -1
    assert h.mri_set_option(b"no_such_option", 1) == -1
    assert h.mri_set_option(b"xcd_affinity", 1) == 0
    with pytest.raises(RuntimeError, match="libmri_inr"):
        lib.call("mri_set_option", b"bogus", 0)
    need = h.mri_hashgrid_backward_workspace_bytes(C.byref(desc), 1 << 10)
    assert need >= (1 << 10) * 8 * 12


def test_every_entry_point_rejects_null_buffers_before_touching_the_device(lib):
    """A non-empty call with NULL buffers (or an impossible range) returns -1 with a message
    from every compute entry point; nothing is launched, so this runs without a GPU."""
    from mri_interpolation_amd import ops
    h = lib.load()
    desc = C.byref(ops.make_grid_desc(3, [[16, 16, 16]], [4096], 2))
    calls = {
        "mri_hashgrid_forward": (desc, None, 8, None, None, 2, 2, 1, None),
        "mri_hashgrid_backward": (desc, None, None, 8, 2, 2, 1, None, 0, None, 0, None),
        "mri_hashgrid_backward_levels": (desc, None, None, 8, 2, 2, 1, None, 0, 1, None, 0, None),
        "mri_hashgrid_backward_prepare": (desc, None, 8, 0, None, 0, None),
        "mri_linear_forward": (None, 4, 1, None, None, 8, 4, 4, 0, 1.0, None, 4, None, 0, None),
        "mri_linear_backward_data": (None, 4, None, 4, 4, 4, 0, None, 4, None, 4, 4, None),
        "mri_linear_backward_weight": (None, 4, None, 4, 4, 4, 4, 4, None, None, None),
        "mri_apply_deriv": (None, 4, 1, None, 4, 4, 4, None),
        "mri_frequency_forward": (None, 3, 8, 3, 4, None, 24, None),
        "mri_frequency_backward": (None, 3, None, 24, 8, 3, 4, None, 3, None),
        "mri_mse_loss": (None, None, 8, 1.0, None, None, None),
        "mri_tiny_mlp_forward": (None, 8, 32, 128, None, None, None, None, None, None, None, None),
        "mri_tiny_mlp_train": (None, None, 8, 32, 128) + (None,) * 6 + (1.0,) + (None,) * 10
                              + (0, None),
        "mri_adam_step": (None, None, None, None, 8, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None),
        "mri_siren_forward": (None, 8, 3, 256, 5, None, None, 30.0, 30.0, None, None, None, None, 0, None),
        "mri_siren_backward": (None, None, 8, 3, 256, 5, None, None, None, None, None, None, 0, None, 0,
                               None),
        "mri_siren_forward_loss": (None, None, 8, 8, 3, 256, 5, None, None, 30.0, 30.0, 1.0, None, None)
                                  + (None,) * 7 + (0, None),
        "mri_sample_indices": (1, 0, 10, 5, 4, None, None),
        "mri_fused_step": (None,),
        "mri_gather_batch": (None, 4, 3, None, None, None, None, None, None, None),
    }
    for name, args in calls.items():
        assert len(args) == len(lib.SIGNATURES[name]), name
        assert getattr(h, name)(*args) == -1, name
        assert h.mri_last_error().decode(), name
    assert h.mri_hashgrid_forward(desc, None, -1, None, None, 2, 2, 1, None) == -1
    assert "out of range" in h.mri_last_error().decode()
    assert h.mri_tiny_mlp_supported(32, 128, 1) == 1 and h.mri_tiny_mlp_supported(64, 128, 1) == 0
    # SIREN workspaces: the forward one holds the split weights (6 H^2 bytes per hidden x hidden layer),
    # the training one the slabs as well; unsupported shapes report -1
    assert h.mri_siren_forward_workspace_bytes(256, 5) == 4 * 6 * 256 * 256
    assert h.mri_siren_forward_workspace_bytes(64, 1) == 0
    assert h.mri_siren_forward_workspace_bytes(96, 3) == -1
    assert h.mri_siren_backward_workspace_bytes(1 << 12, 256, 5) > h.mri_siren_forward_workspace_bytes(256, 5)


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "_LIB_PATH", "/nonexistent/libmri_inr.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        lib.load()
