"""Minimal volume / checkpoint IO for the entry points (SimpleITK/nibabel are not dependencies).

write_nifti: single-file NIfTI-1 (.nii / .nii.gz), enough for the label and CT volumes the reference writes with
SimpleITK (ccdm/ddpm/evaluator.py:147-148, latentdiffusion/sample_diffusion.py:248-250).
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


def load_checkpoint(path: str) -> dict:
    """Safe loader only (executes nothing from the file)."""
    return torch.load(path, map_location="cpu", weights_only=True)
