"""Minimal volume / checkpoint IO for the entry points (SimpleITK/nibabel are not dependencies).

write_nifti / read_nifti: single-file NIfTI-1 (.nii / .nii.gz), enough for the label and CT volumes the reference writes with
SimpleITK (ccdm/ddpm/evaluator.py:147-148, latentdiffusion/sample_diffusion.py:248-250) and reads back with nibabel when the
stage-1 masks are handed to the CT generator (README.md:21; recipe latentdiffusion/sample_diffusion.py:199-200).
Checkpoints are read with torch.load(weights_only=True) only.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np
import torch

_NIFTI_DTYPES = {np.dtype("uint8"): (2, 8), np.dtype("int16"): (4, 16), np.dtype("int32"): (8, 32), np.dtype("float32"): (16, 32)}


def write_nifti(path: str, arr: np.ndarray, spacing=(1.0, 1.0, 1.0)) -> None:
    """arr is [D, H, W] (SimpleITK GetImageFromArray convention: stored as x=W fastest)."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in _NIFTI_DTYPES:
        arr = arr.astype(np.float32)
    code, bits = _NIFTI_DTYPES[arr.dtype]
    D, H, W = arr.shape
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, W, H, D, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, bits)
    struct.pack_into("<8f", hdr, 76, 1.0, spacing[0], spacing[1], spacing[2], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)                 # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)                   # scl_slope
    struct.pack_into("<h", hdr, 254, 1)                     # sform_code
    struct.pack_into("<4f", hdr, 280, spacing[0], 0, 0, 0)
    struct.pack_into("<4f", hdr, 296, 0, spacing[1], 0, 0)
    struct.pack_into("<4f", hdr, 312, 0, 0, spacing[2], 0)
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + arr.tobytes()
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


_NIFTI_CODES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}


def read_nifti(path: str) -> np.ndarray:
    """Inverse of write_nifti: a single-file NIfTI-1 volume (.nii / .nii.gz, little endian) as an array [D, H, W] -- what
    `nibabel.load(path).dataobj[:].transpose(2, 1, 0)` gives (latentdiffusion/sample_diffusion.py:199).  scl_slope / scl_inter are
    applied when they are set to something other than identity (then the result is float32)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352 or struct.unpack_from("<i", raw, 0)[0] != 348:
        raise ValueError(f"{path}: not a little-endian single-file NIfTI-1 volume (sizeof_hdr != 348)")
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: NIfTI-1 magic missing")
    dims = struct.unpack_from("<8h", raw, 40)
    if dims[0] < 3 or any(d != 1 for d in dims[4:1 + dims[0]]):
        raise ValueError(f"{path}: expected a 3-D volume, header dim = {dims}")
    code = struct.unpack_from("<h", raw, 70)[0]
    if code not in _NIFTI_CODES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {code}")
    W, H, D = dims[1], dims[2], dims[3]
    off = int(struct.unpack_from("<f", raw, 108)[0]) or 352
    dt = np.dtype(_NIFTI_CODES[code]).newbyteorder("<")
    n = D * H * W
    if len(raw) < off + n * dt.itemsize:
        raise ValueError(f"{path}: truncated voxel data ({len(raw) - off} bytes for {n} voxels of {dt})")
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(D, H, W)
    slope, inter = struct.unpack_from("<2f", raw, 112)
    if slope not in (0.0, 1.0) or inter != 0.0:
        arr = arr.astype(np.float32) * np.float32(slope) + np.float32(inter)
    return np.ascontiguousarray(arr)


def load_checkpoint(path: str) -> dict:
    """Safe loader only (executes nothing from the file)."""
    return torch.load(path, map_location="cpu", weights_only=True)
