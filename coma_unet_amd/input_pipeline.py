"""Input pipeline of the training path, MI355X side.

The reference loads every sample on the host (`VolumeDataset_ADNI_A4_combined.py:58-133`): SimpleITK reads three NIfTI
volumes, resamples each to 2 mm with a nearest-neighbour `ResampleImageFilter` (identity transform, same origin and
direction), converts to torch, `nan_to_num`s, moves to the GPU and zeroes the MRI where the ROI mask is 0; covariates are
min-max scaled once at dataset construction (:50) and assembled as `[[abeta, age, sex, edu / 30, cog, meta]]` (:85).
Here the raw volumes are copied to the device as they are (pinned host buffers, a side HIP stream) and ONE kernel per
volume does resample + nan_to_num + masking (`coma_resample_nearest`); `Prefetcher` overlaps the next sample's copy and
resample with the current training step.

Parity status: the resampling restates ITK's documented nearest-neighbour semantics (continuous index = output index x
new/old spacing, round half up, default value outside [-0.5, size - 0.5)); SimpleITK is not installed here and the
reference ships no image fixtures, so this is "parity unpinned" against SimpleITK itself (`oracle/input_oracle.py` is the
checker).  NIfTI reading covers single-file NIfTI-1 (.nii / .nii.gz), the format of the reference's `rnu.nii` inputs.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, check, ptr

NEW_SPACING = (2.0, 2.0, 2.0)          # VolumeDataset_ADNI_A4_combined.py:101

# SimpleITK pixel-type ids: the reference passes `volume.GetPixelIDValue()` as the resampler's default pixel value (:121)
SITK_PIXEL_ID = {np.dtype(np.int8): 0, np.dtype(np.uint8): 1, np.dtype(np.int16): 2, np.dtype(np.uint16): 3,
                 np.dtype(np.int32): 4, np.dtype(np.uint32): 5, np.dtype(np.int64): 6, np.dtype(np.uint64): 7,
                 np.dtype(np.float32): 8, np.dtype(np.float64): 9}


def out_size(size_xyz, spacing_xyz, new_spacing=NEW_SPACING):
    """:108-112 -- int(np.round(size * spacing / new_spacing)) per axis (numpy rounds half to even)."""
    return tuple(int(np.round(size_xyz[i] * (spacing_xyz[i] / new_spacing[i]))) for i in range(3))


def resample_nearest(vol: torch.Tensor, spacing_xyz, new_spacing=NEW_SPACING, default_value=0.0, nan_to_num=True,
                     zero_where: torch.Tensor = None, size_xyz=None) -> torch.Tensor:
    """vol: fp32 device tensor (Z, Y, X) as `sitk.GetArrayFromImage` lays it out.  Returns (Zo, Yo, Xo) fp32."""
    assert vol.is_cuda and vol.dtype == torch.float32 and vol.dim() == 3 and vol.is_contiguous()
    Dz, Hy, Wx = vol.shape
    Wo, Ho, Do = size_xyz if size_xyz is not None else out_size((Wx, Hy, Dz), spacing_xyz, new_spacing)
    out = torch.empty((Do, Ho, Wo), dtype=torch.float32, device=vol.device)
    if zero_where is not None:
        assert zero_where.shape == out.shape and zero_where.dtype == torch.float32 and zero_where.is_contiguous()
    check(lib.coma_resample_nearest(ptr(vol), Dz, Hy, Wx, float(spacing_xyz[2]), float(spacing_xyz[1]), float(spacing_xyz[0]),
                                    ptr(out), Do, Ho, Wo, float(new_spacing[2]), float(new_spacing[1]), float(new_spacing[0]),
                                    float(default_value), int(bool(nan_to_num)), ptr(zero_where), L.stream()),
          "coma_resample_nearest")
    return out


def prepare_sample(mri, tau, roi, spacing_xyz, new_spacing=NEW_SPACING, default_values=(8.0, 8.0, 8.0), resize=True):
    """`__getitem__` of :58-66 for three raw (Z, Y, X) fp32 device volumes of one subject: returns
    (mri, tau, roi) as (1, D, H, W) fp32 tensors, the MRI zeroed where the (resampled) ROI mask is 0.
    default_values: the SimpleITK pixel ids of the three files (float32 images: 8), see SITK_PIXEL_ID."""
    if not resize:
        roi_r = torch.nan_to_num(roi)
        mri_r = torch.nan_to_num(mri) * (roi_r != 0)
        return mri_r.unsqueeze(0), torch.nan_to_num(tau).unsqueeze(0), roi_r.unsqueeze(0)
    roi_r = resample_nearest(roi, spacing_xyz, new_spacing, default_values[2])
    tau_r = resample_nearest(tau, spacing_xyz, new_spacing, default_values[1])
    mri_r = resample_nearest(mri, spacing_xyz, new_spacing, default_values[0], zero_where=roi_r)
    return mri_r.unsqueeze(0), tau_r.unsqueeze(0), roi_r.unsqueeze(0)


def covariate_row(abeta, age, sex, edu, cog, meta):
    """:85 -- (1, 6) float64: education is divided by 30, the rest is passed as stored in the (already scaled) tables."""
    return torch.from_numpy(np.array([[abeta, age, sex, edu / 30, cog, meta]], dtype=np.float64))


def minmax_scale(columns: np.ndarray) -> np.ndarray:
    """:50 -- (x - min) / (max - min) per column of the covariate table."""
    columns = np.asarray(columns, dtype=np.float64)
    lo, hi = np.nanmin(columns, axis=0), np.nanmax(columns, axis=0)
    return (columns - lo) / (hi - lo)


# ---------------------------------------------------------------------------------------------------------------------
# minimal NIfTI-1 reader (single file .nii / .nii.gz): array in (Z, Y, X) order + (x, y, z) spacing, like
# sitk.GetArrayFromImage / GetSpacing.  scl_slope / scl_inter are applied when set.
# ---------------------------------------------------------------------------------------------------------------------
_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                 768: np.uint32, 1024: np.int64, 1280: np.uint64}


def read_nifti(path):
    """-> (float32 array (Z, Y, X), spacing (sx, sy, sz), stored numpy dtype)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    for endian in ("<", ">"):
        if struct.unpack(endian + "i", raw[0:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    dim = struct.unpack(endian + "8h", raw[40:56])
    datatype = struct.unpack(endian + "h", raw[70:72])[0]
    pixdim = struct.unpack(endian + "8f", raw[76:108])
    vox_offset = int(struct.unpack(endian + "f", raw[108:112])[0])
    slope, inter = struct.unpack(endian + "2f", raw[112:120])
    if raw[344:347] not in (b"n+1",):
        raise ValueError(f"{path}: only single-file NIfTI-1 (magic n+1) is supported")
    if datatype not in _NIFTI_DTYPES or dim[0] < 3:
        raise ValueError(f"{path}: unsupported datatype {datatype} / rank {dim[0]}")
    nx, ny, nz = dim[1], dim[2], dim[3]
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(endian)
    arr = np.frombuffer(raw, dtype=dt, count=nx * ny * nz, offset=vox_offset).reshape(nz, ny, nx)
    out = arr.astype(np.float32)
    if slope not in (0.0,) and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope):
        out = out * np.float32(slope) + np.float32(inter)
    return np.ascontiguousarray(out), (float(pixdim[1]), float(pixdim[2]), float(pixdim[3])), np.dtype(_NIFTI_DTYPES[datatype])


class Prefetcher:
    """Wraps an iterable of host samples {'mri','tau','roi': (Z,Y,X) float32 numpy arrays, 'spacing': (sx,sy,sz), optional
    other keys passed through} and yields device samples {'mri','tau','roi': (1,D,H,W) fp32 on `device`, ...}: the next
    sample's host->device copies (pinned staging buffers) and its resample kernels run on a side stream while the caller
    trains on the current one."""

    def __init__(self, samples, device="cuda", new_spacing=NEW_SPACING, resize=True):
        self.it = iter(samples)
        self.device = torch.device(device)
        self.new_spacing, self.resize = new_spacing, resize
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = {}
        self._next = None
        self._fetch()

    def _to_device(self, key, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        buf = self._pinned.get(key)
        if buf is None or buf.shape != arr.shape:
            buf = torch.empty(arr.shape, dtype=torch.float32).pin_memory()
            self._pinned[key] = buf
        buf.copy_(torch.from_numpy(arr))
        return buf.to(self.device, non_blocking=True)

    def _fetch(self):
        try:
            s = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            # the pinned staging buffers are reused: the previous copies were waited for by the consumer's wait_stream
            vols = {k: self._to_device(k, s[k]) for k in ("mri", "tau", "roi")}
            dv = tuple(float(SITK_PIXEL_ID.get(np.dtype(s.get("stored_dtype", np.float32)), 8)) for _ in range(3))
            mri, tau, roi = prepare_sample(vols["mri"], vols["tau"], vols["roi"], s["spacing"], self.new_spacing, dv, self.resize)
            out = {k: v for k, v in s.items() if k not in ("mri", "tau", "roi")}
            out.update(mri=mri, tau=tau, roi=roi)
        self._next = out

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        cur = self._next
        for k in ("mri", "tau", "roi"):
            cur[k].record_stream(torch.cuda.current_stream(self.device))
        self.stream.wait_stream(torch.cuda.current_stream(self.device))     # staging buffers free again
        self._fetch()
        return cur
