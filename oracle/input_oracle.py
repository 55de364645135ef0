"""TEST INFRASTRUCTURE (not the product path): numpy restatement of the reference's per-sample input pipeline
(VolumeDataset_ADNI_A4_combined.py:58-133) -- SimpleITK nearest-neighbour resample to 2 mm + nan_to_num + MRI masking.

PARITY UNPINNED: SimpleITK is not installed in this image and the reference holds no image fixtures, so this restates
ITK's documented behaviour (ResampleImageFilter with an identity transform and the input's origin / direction:
continuous input index = output index * new_spacing / old_spacing; NearestNeighborInterpolateImageFunction rounds with
itk::Math::RoundHalfIntegerUp; samples outside [-0.5, size - 0.5) take the default pixel value) rather than its output.
"""
import numpy as np


def out_size(size_xyz, spacing_xyz, new_spacing=(2.0, 2.0, 2.0)):
    return tuple(int(np.round(size_xyz[i] * (spacing_xyz[i] / new_spacing[i]))) for i in range(3))    # :108-112


def nn_indices(n_out, ratio, size):
    c = np.arange(n_out, dtype=np.float64) * np.float64(ratio)
    inside = (c >= -0.5) & (c < size - 0.5)
    idx = np.floor(c + 0.5).astype(np.int64)
    return np.where(inside, idx, -1)


def resample_nearest(vol_zyx, spacing_xyz, new_spacing=(2.0, 2.0, 2.0), default_value=0.0, nan_to_num=True,
                     zero_where=None):
    vol = np.asarray(vol_zyx, dtype=np.float32)
    Dz, Hy, Wx = vol.shape
    Wo, Ho, Do = out_size((Wx, Hy, Dz), spacing_xyz, new_spacing)
    iz = nn_indices(Do, new_spacing[2] / spacing_xyz[2], Dz)
    iy = nn_indices(Ho, new_spacing[1] / spacing_xyz[1], Hy)
    ix = nn_indices(Wo, new_spacing[0] / spacing_xyz[0], Wx)
    out = np.full((Do, Ho, Wo), np.float32(default_value), dtype=np.float32)
    vz, vy, vx = iz >= 0, iy >= 0, ix >= 0
    sub = vol[np.ix_(iz[vz], iy[vy], ix[vx])]
    out[np.ix_(np.nonzero(vz)[0], np.nonzero(vy)[0], np.nonzero(vx)[0])] = sub
    if nan_to_num:
        out = np.nan_to_num(out, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min)   # torch defaults
    if zero_where is not None:
        out = np.where(zero_where == 0, np.float32(0), out)
    return out.astype(np.float32)


def prepare_sample(mri, tau, roi, spacing_xyz, new_spacing=(2.0, 2.0, 2.0), default_values=(8.0, 8.0, 8.0)):
    roi_r = resample_nearest(roi, spacing_xyz, new_spacing, default_values[2])
    tau_r = resample_nearest(tau, spacing_xyz, new_spacing, default_values[1])
    mri_r = resample_nearest(mri, spacing_xyz, new_spacing, default_values[0])
    mri_r = np.where(roi_r == 0, np.float32(0), mri_r)                       # :62
    return mri_r[None], tau_r[None], roi_r[None]
