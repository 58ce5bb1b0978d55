"""Minimal NIfTI-1 reader / writer (single-file .nii / .nii.gz), NumPy only.

Stands in for the two nibabel calls on the reference's path: `nib.load(path).get_fdata(
dtype=np.float32)` (reference datamodules.py:135-138) and `nib.save(nib.Nifti1Image(im,
affine=np.eye(4)), path)` (reference launcher.py:189,219-222).  nibabel is not installed on
the MI355X image.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
           256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _read_all(path):
    with open(path, "rb") as f:
        head = f.read(2)
    opener = gzip.open if head == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        return f.read()


def read_header(path):
    raw = _read_all(path)[:352]
    return _parse_header(raw)[0]


def _parse_header(raw):
    for endian in ("<", ">"):
        if struct.unpack(endian + "i", raw[:4])[0] == 348:
            break
    else:
        raise ValueError("not a NIfTI-1 file (sizeof_hdr != 348)")
    dims = struct.unpack(endian + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(endian + "2h", raw[70:74])
    vox_offset = struct.unpack(endian + "f", raw[108:112])[0]
    slope, inter = struct.unpack(endian + "2f", raw[112:120])
    if raw[344:348] not in (b"n+1\x00", b"ni1\x00"):
        raise ValueError("bad NIfTI-1 magic")
    hdr = dict(shape=tuple(int(d) for d in dims[1:1 + dims[0]]), datatype=datatype,
               bitpix=bitpix, vox_offset=int(vox_offset), scl_slope=slope, scl_inter=inter)
    return hdr, endian


def load(path) -> np.ndarray:
    """Voxel array as float32 with scl_slope/scl_inter applied (`get_fdata(dtype=float32)`)."""
    raw = _read_all(path)
    hdr, endian = _parse_header(raw[:352])
    if hdr["datatype"] not in _DTYPES:
        raise ValueError(f"unsupported NIfTI datatype {hdr['datatype']}")
    dt = np.dtype(_DTYPES[hdr["datatype"]]).newbyteorder(endian)
    count = int(np.prod(hdr["shape"]))
    data = np.frombuffer(raw, dtype=dt, count=count, offset=hdr["vox_offset"])
    data = data.reshape(hdr["shape"], order="F")  # NIfTI stores the first axis fastest
    slope, inter = hdr["scl_slope"], hdr["scl_inter"]
    if slope == 0 or not np.isfinite(slope):
        slope, inter = 1.0, 0.0
    # nibabel scales in float64, then casts to the requested dtype
    return (data.astype(np.float64) * slope + inter).astype(np.float32)


def save(array: np.ndarray, path, affine=None, scl_slope: float = 1.0, scl_inter: float = 0.0):
    """Write `array` with an identity affine unless one is given.  `scl_slope` / `scl_inter` go
    into the header as they are (the stored voxels are NOT rescaled): integer voxels plus a slope
    is how scanners -- and the reference's sample volume -- store intensities."""
    arr = np.asarray(array)
    if arr.dtype.name not in _CODES:
        arr = arr.astype(np.float32)
    if arr.ndim > 7:
        raise ValueError("NIfTI-1 holds at most 7 dimensions")
    aff = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dims = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dims)
    struct.pack_into("<2h", hdr, 70, _CODES[arr.dtype.name], arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, *([1.0] * 7))      # pixdim
    struct.pack_into("<f", hdr, 108, 352.0)                   # vox_offset
    struct.pack_into("<2f", hdr, 112, float(scl_slope), float(scl_inter))
    struct.pack_into("<h", hdr, 254, 2)                       # sform_code = aligned
    struct.pack_into("<4f", hdr, 280, *aff[0])
    struct.pack_into("<4f", hdr, 296, *aff[1])
    struct.pack_into("<4f", hdr, 312, *aff[2])
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + b"\x00" * 4 + arr.astype(arr.dtype.newbyteorder("<")).tobytes(order="F")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(payload)
