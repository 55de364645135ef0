"""the input pipeline (resample to 2 mm + nan_to_num + masking, NIfTI-1 reading, covariates)."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch


def _write_nifti(path, arr_zyx, spacing, datatype=16, slope=0.0, inter=0.0):
    nz, ny, nx = arr_zyx.shape
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, datatype)
    struct.pack_into("<h", hdr, 72, {2: 8, 4: 16, 16: 32, 64: 64}[datatype])
    struct.pack_into("<8f", hdr, 76, 1.0, spacing[0], spacing[1], spacing[2], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, slope, inter)
    hdr[344:348] = b"n+1\x00"
    np_dt = {2: np.uint8, 4: np.int16, 16: np.float32, 64: np.float64}[datatype]
    data = bytes(hdr) + np.ascontiguousarray(arr_zyx.astype(np_dt)).tobytes()
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(data)


def test_oracle_resample_known_cases():
    from oracle import input_oracle as O
    v = np.arange(4 * 6 * 8, dtype=np.float32).reshape(4, 6, 8)
    # 1 mm -> 2 mm: every second voxel, exactly
    r = O.resample_nearest(v, (1.0, 1.0, 1.0))
    assert r.shape == (2, 3, 4) and np.array_equal(r, v[::2, ::2, ::2])
    # same spacing: identity
    assert np.array_equal(O.resample_nearest(v, (2.0, 2.0, 2.0)), v)
    # 3 mm -> 2 mm (upsampling): size round(8*1.5)=12; index o -> floor(o*2/3 + .5); past the volume -> default value
    r = O.resample_nearest(v, (3.0, 3.0, 3.0), default_value=8.0)
    assert r.shape == (6, 9, 12)
    assert [int(x) for x in r[0, 0, :]] == [0, 1, 1, 2, 3, 3, 4, 5, 5, 6, 7, 7]
    # rounding of the output size is numpy's (half to even): 5 voxels of 1 mm -> round(2.5) = 2
    assert O.out_size((5, 5, 5), (1.0, 1.0, 1.0)) == (2, 2, 2)
    # a sample beyond size - 0.5 takes the default: 5 voxels of 2.5 mm -> round(6.25) = 6 outputs, o=5 -> c=4.0 inside;
    # 4 voxels of 2.6 mm -> round(5.2)=5 outputs, o=4 -> c=3.077 inside; force it with 3 voxels of 2.9 mm: round(4.35)=4, o=3 -> 2.07
    w = np.arange(3, dtype=np.float32).reshape(1, 1, 3)
    assert [int(x) for x in O.resample_nearest(w, (2.9, 2.0, 2.0), default_value=-7.0)[0, 0]] == [0, 1, 1, 2]
    w2 = np.arange(2, dtype=np.float32).reshape(1, 1, 2)        # 2 voxels of 3.4 mm -> round(3.4)=3 outputs; o=2 -> c=1.18 inside
    assert O.resample_nearest(w2, (3.4, 2.0, 2.0)).shape == (1, 1, 3)
    # nan_to_num + masking
    v2 = v.copy(); v2[0, 0, 0] = np.nan; v2[0, 0, 2] = np.inf
    r = O.resample_nearest(v2, (1.0, 1.0, 1.0))
    assert r[0, 0, 0] == 0.0 and r[0, 0, 1] == np.finfo(np.float32).max
    roi = (v % 3 == 0).astype(np.float32)
    m, t, ro = O.prepare_sample(v, v + 1, roi, (1.0, 1.0, 1.0))
    assert m.shape == (1, 2, 3, 4) and np.all(m[ro == 0] == 0) and np.array_equal(t[0], (v + 1)[::2, ::2, ::2])


def test_nifti_reader_and_covariates(tmp_path):
    from coma_unet_amd import input_pipeline as P     # (imports the HIP library: present after build())
    rng = np.random.default_rng(0)
    a = rng.normal(size=(5, 6, 7)).astype(np.float32)
    for name in ("a.nii", "b.nii.gz"):
        _write_nifti(tmp_path / name, a, (1.0, 1.2, 1.5))
        arr, sp, dt = P.read_nifti(str(tmp_path / name))
        assert np.array_equal(arr, a) and sp == (1.0, pytest.approx(1.2), 1.5) and dt == np.float32
    b = rng.integers(-100, 100, size=(3, 4, 5)).astype(np.int16)
    _write_nifti(tmp_path / "c.nii", b, (2.0, 2.0, 2.0), datatype=4, slope=0.5, inter=1.0)
    arr, sp, dt = P.read_nifti(str(tmp_path / "c.nii"))
    assert np.allclose(arr, b.astype(np.float32) * 0.5 + 1.0) and dt == np.int16 and P.SITK_PIXEL_ID[dt] == 2
    with pytest.raises(ValueError):
        (tmp_path / "bad.nii").write_bytes(b"\x00" * 400)
        P.read_nifti(str(tmp_path / "bad.nii"))
    row = P.covariate_row(1.0, 0.4, 1.0, 15.0, 0.3, 1.2)
    assert row.dtype == torch.float64 and row.shape == (1, 6) and float(row[0, 3]) == 0.5
    t = np.array([[1.0, 10.0], [3.0, 30.0], [2.0, 20.0]])
    assert np.allclose(P.minmax_scale(t), [[0, 0], [1, 1], [0.5, 0.5]])
    assert P.out_size((256, 256, 170), (1.0, 1.0, 1.5)) == (128, 128, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,spacing", [((37, 41, 53), (1.0, 1.0, 1.0)), ((20, 24, 28), (2.0, 2.0, 2.0)),
                                           ((30, 33, 35), (1.2, 0.9375, 1.5)), ((9, 10, 11), (3.0, 3.4, 2.9)),
                                           ((170, 256, 256), (1.0, 1.0, 1.5))])
def test_resample_kernel_bit_exact_vs_oracle(shape, spacing):
    from coma_unet_amd import input_pipeline as P
    from oracle import input_oracle as O
    rng = np.random.default_rng(sum(shape))
    v = rng.normal(size=shape).astype(np.float32)
    v[0, 0, 0] = np.nan; v[-1, -1, -1] = np.inf; v[1, 2, 3] = -np.inf
    roi = (rng.random(size=shape) > 0.3).astype(np.float32) * rng.integers(1, 2036, size=shape).astype(np.float32)
    dv = torch.from_numpy(v).cuda()
    got = P.resample_nearest(dv, spacing, default_value=8.0).cpu().numpy()
    want = O.resample_nearest(v, spacing, default_value=8.0)
    assert got.shape == want.shape and np.array_equal(got, want)
    m, t, r = P.prepare_sample(dv, dv + 1, torch.from_numpy(roi).cuda(), spacing)
    om, ot, orr = O.prepare_sample(v, v + 1, roi, spacing)
    assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(t.cpu().numpy(), ot) and np.array_equal(r.cpu().numpy(), orr)


@pytest.mark.gpu
def test_prefetcher_yields_prepared_samples():
    from coma_unet_amd import input_pipeline as P
    from oracle import input_oracle as O
    rng = np.random.default_rng(3)
    samples = []
    for i in range(3):
        shp = (24 + 2 * i, 20, 22)
        samples.append({"mri": rng.random(shp).astype(np.float32), "tau": rng.random(shp).astype(np.float32),
                        "roi": (rng.random(shp) > 0.5).astype(np.float32), "spacing": (1.0, 1.0, 1.0), "id": i})
    got = list(P.Prefetcher(samples))
    assert [g["id"] for g in got] == [0, 1, 2]
    torch.cuda.synchronize()
    for s, g in zip(samples, got):
        om, ot, orr = O.prepare_sample(s["mri"], s["tau"], s["roi"], s["spacing"])
        assert np.array_equal(g["mri"].cpu().numpy(), om) and np.array_equal(g["tau"].cpu().numpy(), ot)
        assert np.array_equal(g["roi"].cpu().numpy(), orr)
