"""Training configuration with the reference's field names and default VALUES.

Mirrors reference `config/base.py`: `BaseConfig` (:17-55) and `HashConfig` (:57-89).  The
reference evaluates `nib.load(image_path).shape` at class-definition time (config/base.py:22,61)
and binds model / datamodule classes into the dataclass; here `image_shape` and `dim_in`
are resolved lazily by `resolve()` and classes are looked up by name (launcher `--model_class`).
`config/hash_config.json` is read into `enco_config` as the reference does (launcher.py:73-74);
`encoder_from_json()` maps its tiny-cuda-nn style keys onto the Python encoder
(per_level_scale -> finest_resolution = base * scale**(L-1)).
"""
import json
import os
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple, Union


@dataclass
class BaseConfig:
    checkpoint_path: Optional[str] = None
    image_path: str = "sample_ankle_dyn_mri.nii.gz"
    image_shape: Optional[Tuple[int, ...]] = None   # filled by resolve()
    batch_size: int = 4096
    epochs: int = 1
    num_workers: int = 0            # batches are generated on the GPU; kept for CLI parity
    accumulate_grad_batches: Any = None
    seed: int = 1337                # reference launcher.py:30
    # network parameters
    dim_in: Optional[int] = None    # len(image_shape), filled by resolve()
    dim_hidden: int = 128
    dim_out: int = 1
    n_layers: int = 6
    n_sample: int = 3
    w0: float = 30.0
    w0_initial: float = 30.0
    use_bias: bool = True
    final_activation: Any = None
    lr: float = 1e-4
    model_class: str = "SirenNet"
    norm_siren: bool = False
    interp_shapes: List[Tuple[int, ...]] = field(default_factory=list)
    slice_spec: Optional[str] = None  # e.g. ":,:,3,7" -> train on a sub-volume
    enco_config: Optional[dict] = None
    log: str = ""

    def resolve(self, volume_shape: Sequence[int]):
        self.image_shape = tuple(int(s) for s in volume_shape)
        self.dim_in = len(self.image_shape)
        return self

    def export_to_txt(self, file_path: str = "") -> None:
        """`config.txt`, one `key : value` line per field (reference config/base.py:52-55)."""
        with open(os.path.join(file_path, "config.txt"), "w") as f:
            for key, value in self.__dict__.items():
                f.write(f"{key} : {value}\n")


@dataclass
class HashConfig(BaseConfig):
    interp_shapes: List[Tuple[int, ...]] = field(default_factory=lambda: [(352, 352, 30)])
    batch_size: int = 10000
    encoder_type: str = "hash"
    n_levels: int = 4
    n_features_per_level: int = 1
    log2_hashmap_size: int = 23
    base_resolution: Union[int, Tuple[int, ...]] = (64, 64, 5)
    finest_resolution: Union[int, Tuple[int, ...]] = (352, 352, 15)
    per_level_scale: float = 1.2
    interpolation: str = "Linear"
    dim_hidden: int = 64
    n_layers: int = 2
    lr: float = 5e-3
    dropout: float = 0.0
    model_class: str = "HashMLP"
    # decoder of the reference (Linear -> BatchNorm1d -> GELU -> Dropout) by default; the fused
    # MI355X tiny-MLP is selected with activation="ReLU", batch_norm=False, final_activation_on=False
    activation: str = "GELU"
    batch_norm: bool = True
    final_activation_on: bool = True


def load_json(path: str) -> dict:
    with open(path) as f:
        return json.load(f)


def encoder_from_json(cfg: dict, dim: int) -> dict:
    """Keyword arguments for `MultiResHashGrid` from a tiny-cuda-nn style `encoding` block
    (reference config/hash_config.json:13-20)."""
    enc = cfg.get("encoding", cfg)
    base = int(enc.get("base_resolution", 16))
    levels = int(enc.get("n_levels", 16))
    scale = float(enc.get("per_level_scale", 2.0))
    return dict(dim=dim, n_levels=levels,
                n_features_per_level=int(enc.get("n_features_per_level", 2)),
                log2_hashmap_size=int(enc.get("log2_hashmap_size", 19)),
                base_resolution=base, finest_resolution=base * scale ** (levels - 1))


def apply_overrides(config, overrides: dict):
    """Copy parsed CLI arguments into the config by name (reference launcher.py:77-80)."""
    for key, value in overrides.items():
        if value is not None:
            setattr(config, key, value)
    return config


def parse_slice(spec: str):
    """':,:,3,7' -> (slice(None), slice(None), 3, 7)."""
    out = []
    for tok in spec.split(","):
        tok = tok.strip()
        if ":" in tok:
            parts = [int(p) if p else None for p in tok.split(":")]
            out.append(slice(*parts))
        else:
            out.append(int(tok))
    return tuple(out)


def parse_resolution(spec: str):
    """'16' -> 16 (isotropic grid), '16,16,5,7' -> (16.0, 16.0, 5.0, 7.0) (per-axis grid)."""
    vals = [float(t) for t in spec.split(",")]
    if len(vals) == 1:
        return int(vals[0]) if vals[0] == int(vals[0]) else vals[0]
    return tuple(vals)
