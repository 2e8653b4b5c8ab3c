#!/usr/bin/env python3
"""Generates tests/golden/twin_ps{32,48,64}.npz: values produced by ATen itself, for the arithmetic that IS in the reference tree.

PROVENANCE.  The reference (oxabz/nuclei-feature-extraction) is Rust over `tch` -> libtorch ATen; it cannot be built here and ships
no fixtures.  What can be run here is ATen: tests/_torch_twin.py issues, op for op, the sequences of shape.rs:137-226 (centre of
mass, covariance -> linalg_eig -> axes / row-of-V orientation, eccentricity, elliptic deviation), color.rs:117-155 (mean_std, the
batch-coupled circular_mean, the in-place hue shift) and texture.rs:332-344 (gray, masked mean / variance of the filter
responses) on torch-CPU -- the same ATen CPU kernels and MKL LAPACK the reference's libtorch-CPU path reaches.  This script runs
that twin on seeded synthetic batches and stores its OUTPUTS next to the INPUTS, so that the `-m gpu` test
(tests/test_twin_golden_gpu.py) can compare the HIP trait entry with ATen-produced numbers directly -- no oracle in between.

What is NOT ATen here (dependency-resident functions whose sources are absent): the inputs those functions would
hand to the in-tree arithmetic are produced by this repository's restatements and stored as data --
  masks        tch_utils::shapes::polygon   -> oracle rasteriser
  ellipse mask tch_utils::shapes::ellipse   -> oracle rasteriser, from the TWIN's centre / axes / angle
  hsv, hed     tch_utils::color::*          -> the torch statement of the build decision (_torch_twin.*_spec)
  filter bank  tch_utils::gabor             -> the oracle's 48 x 30 x 30 kernels, applied by ATen conv2d (f32)
so the fixtures pin S2-S4, S6, C2, C3 and B1's statistics given those inputs; they do not pin the inputs.

    python tests/golden/make_twin_golden.py        (build container only: needs torch-CPU; the GPU box only reads the .npz)
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc        # noqa: E402  (dependency-function stand-ins only: rasterisers, filter bank)
import _parity as par        # noqa: E402
import _torch_twin as tw     # noqa: E402

torch.set_num_threads(1)
N, BATCH = 24, 12            # two reference batches: the hue mean couples the nuclei of a batch (color.rs:144-155)
SEED = 91


def shape_columns(mask, ps):
    """shape.rs:66-111 for one nucleus, ATen ops: (major, minor, eccentricity, orientation, eliptic_deviation, cov p q r, ellipse bits)"""
    m = torch.from_numpy(mask).unsqueeze(0)
    res = tw.major_minor_axes_w_angle(m)
    major, minor, angle = res[0], res[1], res[2]
    if math.isnan(major):
        return [math.nan] * 5, [math.nan] * 3, np.zeros((ps, ps), np.float32)
    cov = res[3]
    com = tw.center_of_mass(m)                                   # (row, col)
    half = np.float32(ps) / np.float32(2.0)                      # shape.rs:71-73: f32 arithmetic
    yc, xc = np.float32(com[0]) - half, np.float32(com[1]) - half  # shape.rs:83: (x_c, y_c) = (centroid[1], centroid[0])
    ell = orc.raster_ellipse(ps, xc, yc, major, minor, angle)
    ed = tw.eliptic_deviation(m, torch.from_numpy(ell).unsqueeze(0))
    return [major, minor, tw.eccentricity(major, minor), angle, ed], [float(cov[0, 0]), float(cov[0, 1]), float(cov[1, 1])], ell


def build(ps):
    off, xy, pu8 = par.make_inputs(SEED, N, ps)
    n = len(off) - 1
    centred = np.concatenate([orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1] for i in range(n)]).astype(np.float32)
    masks = np.stack([orc.raster_polygon(centred[off[i]:off[i + 1]], ps) for i in range(n)])
    patches = pu8.astype(np.float32) / np.float32(255.0)          # utils.rs:172
    shape, cov, ell_bits = [], [], []
    for i in range(n):
        s, c, ell = shape_columns(masks[i], ps)
        shape.append(s); cov.append(c); ell_bits.append(ell)
    color = np.zeros((n, 18), np.float32)
    gabor = np.zeros((n, 96), np.float32)
    bank = torch.from_numpy(orc.gabor_bank()).unsqueeze(1)        # [48,1,30,30]
    for b0 in range(0, n, BATCH):
        p = torch.from_numpy(patches[b0:b0 + BATCH])
        m = torch.from_numpy(masks[b0:b0 + BATCH]).unsqueeze(1)
        color[b0:b0 + BATCH] = tw.color_features(p, m, tw.hsv_from_rgb_spec(p), tw.hed_from_rgb_spec(p)).numpy()
        gs = tw.gray(p)                                           # texture.rs:332
        filtered = torch.nn.functional.conv2d(torch.nn.functional.pad(gs, (15, 14, 15, 14)), bank)   # ATen f32 conv2d
        gabor[b0:b0 + BATCH] = tw.gabor_stats(filtered, m).numpy()  # texture.rs:336-344
    return dict(ps=np.int32(ps), batch_size=np.int32(BATCH), poly_offsets=off.astype(np.uint32), poly_xy=centred, patches_u8=pu8,
                mask_bits=np.packbits(masks.astype(bool), axis=-1), ellipse_bits=np.packbits(np.stack(ell_bits).astype(bool), axis=-1),
                twin_shape=np.asarray(shape, np.float32), twin_cov=np.asarray(cov, np.float32), twin_color=color, twin_gabor=gabor,
                torch_version=np.bytes_(torch.__version__.encode()))


if __name__ == "__main__":
    for ps in (32, 48, 64):
        d = build(ps)
        path = os.path.join(HERE, f"twin_ps{ps}.npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path), "bytes")
