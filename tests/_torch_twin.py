"""torch-CPU twin of the reference's IN-TREE arithmetic (test infrastructure).

Each function issues the same ATen op sequence the reference issues through the `tch` crate
(tch 0.11 -> libtorch ATen; here torch-CPU -> the same ATen CPU kernels and MKL LAPACK), so the
oracle's restatement of src/features/*.rs can be pinned to what libtorch really computes.
Dependency-resident functions (tch_utils::*, geometric_features::*) are NOT twinned: their
sources are absent; inputs they would produce are passed in.
"""
import math

import torch


def center_of_mass(mask):                       # shape.rs:219-226 ; mask [1,ps,ps]
    nz = mask.nonzero()[:, -2:]
    centroid = nz.mean(dim=[0], keepdim=False, dtype=torch.float32)
    return [float(centroid[0].double()), float(centroid[1].double())]


def major_minor_axes_w_angle(mask):             # shape.rs:141-203
    nan = float("nan")
    nz = mask.squeeze().nonzero()
    centroid = nz.mean(dim=[0], keepdim=False, dtype=torch.float32)
    if centroid.size(0) == 0:
        return nan, nan, nan
    points = nz - centroid
    cov = points.transpose(0, 1).mm(points) / points.size(0)
    if bool(cov.isnan().any()):
        return nan, nan, nan
    if cov.size(0) != 2 or cov.size(1) != 2:
        return nan, nan, nan
    if bool(cov.isinf().any()):
        return nan, nan, nan
    eigenvalues, eigenvector = torch.linalg.eig(cov)
    f32 = lambda t: float(torch.tensor(float(t.real.double()), dtype=torch.float32))
    a = f32(eigenvalues[0])
    b = f32(eigenvalues[1])
    if a > b:
        major, minor, mc = math.sqrt(a) if a >= 0 else nan, math.sqrt(b) if b >= 0 else nan, eigenvector[0]
    else:
        major, minor, mc = math.sqrt(b) if b >= 0 else nan, math.sqrt(a) if a >= 0 else nan, eigenvector[1]
    to32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    major, minor = to32(major), to32(minor)
    x = f32(mc[0])
    y = f32(mc[1])
    angle = float(torch.atan2(torch.tensor(x, dtype=torch.float32), torch.tensor(y, dtype=torch.float32)))
    return to32(major * 2.0), to32(minor * 2.0), angle, cov


def eccentricity(major, minor):                 # shape.rs:205-207
    M = torch.tensor(major, dtype=torch.float32) * 0.5
    m = torch.tensor(minor, dtype=torch.float32) * 0.5
    return float((M * M - m * m).sqrt() / M)


def eliptic_deviation(mask, ellipse_mask):      # shape.rs:209-217
    mask = mask.to(torch.float32)
    ellipse_mask = ellipse_mask.to(torch.float32)
    mask_area = float(mask.sum(dtype=torch.float32))
    delta = mask - ellipse_mask
    return float((delta.abs().sum(dtype=torch.float32) / mask_area).double())


def mean_std(img, mask):                        # color.rs:117-134
    masked = img * mask
    mask_area = mask.sum(dim=[-1, -2], keepdim=True, dtype=torch.float32)
    mean = masked.sum(dim=[-1, -2], keepdim=True, dtype=torch.float32)
    mean /= mask_area
    std = img - mean
    std = std.square() * mask
    std = std.sum(dim=[-1, -2], keepdim=True, dtype=torch.float32)
    std /= mask_area
    std = std.sqrt()
    std.squeeze_()
    mean.squeeze_()
    return mean, std


def circular_mean(image, mask):                 # color.rs:144-155 ; image [N,ps,ps], mask [N,1,ps,ps]
    mask_area = mask.sum(dim=[-1, -2, -3], keepdim=False, dtype=torch.float32)
    img = image.deg2rad()
    cos = img.cos() * mask                       # broadcasts to [N,N,ps,ps]
    sin = img.sin() * mask
    cos = cos.sum(dim=[-1, -2, -3], keepdim=False, dtype=torch.float32) / mask_area
    sin = sin.sum(dim=[-1, -2, -3], keepdim=False, dtype=torch.float32) / mask_area
    return (sin.atan2(cos).rad2deg_() + 360.0).fmod_(360.0)


def color_features(patchs, masks, hsv, hed):    # color.rs:43-99 ; hsv/hed = dep outputs
    hsv = hsv.clone()
    mean_rgb, std_rgb = mean_std(patchs, masks)
    mean_hed, std_hed = mean_std(hed, masks)
    h = hsv.select(-3, 0)
    mean_h = circular_mean(h, masks)
    h -= mean_h.view([-1, 1, 1])                # mutates hsv through the view
    mean_hsv, std_hsv = mean_std(hsv, masks)
    if mean_rgb.dim() == 1:                     # N == 1: squeeze_ dropped the batch dim too
        mean_rgb, std_rgb, mean_hed, std_hed, mean_hsv, std_hsv = [t.unsqueeze(0) for t in (mean_rgb, std_rgb, mean_hed, std_hed, mean_hsv, std_hsv)]
    cols = [mean_rgb[:, 0], mean_rgb[:, 1], mean_rgb[:, 2], std_rgb[:, 0], std_rgb[:, 1], std_rgb[:, 2],
            mean_h, mean_hsv[:, 1], mean_hsv[:, 2], std_hsv[:, 0], std_hsv[:, 1], std_hsv[:, 2],
            mean_hed[:, 0], mean_hed[:, 1], mean_hed[:, 2], std_hed[:, 0], std_hed[:, 1], std_hed[:, 2]]
    return torch.stack(cols, dim=1)


def gray(patchs):                               # texture.rs:36,189,332
    return patchs.mean(dim=[-3], keepdim=True, dtype=torch.float32)


def gabor_stats(filtered, masks):               # texture.rs:336-344 ; filtered [N,48,ps,ps]
    n, fc = filtered.shape[0], filtered.shape[1]
    masks_areas = masks.sum(dim=[-3, -2, -1])
    mean = (filtered * masks).sum(dim=[-2, -1]) / masks_areas.unsqueeze(-1)
    variance = ((filtered - mean.view([n, fc, 1, 1])).square() * masks).sum(dim=[-2, -1]) / masks_areas.unsqueeze(-1)
    out = torch.empty(n, 2 * fc)
    out[:, 0::2] = mean
    out[:, 1::2] = variance
    return out


# ---- torch restatement of the spec'd dep functions used only to feed the twins above ----------
def hsv_from_rgb_spec(p):                       # same definition as oracle orc_hsv
    r, g, b = p[:, 0], p[:, 1], p[:, 2]
    mx = torch.maximum(r, torch.maximum(g, b))
    mn = torch.minimum(r, torch.minimum(g, b))
    d = mx - mn
    safe = torch.where(d == 0, torch.ones_like(d), d)
    hr = torch.fmod((g - b) / safe, 6.0)
    hr = torch.where(hr < 0, hr + 6.0, hr)
    hg = (b - r) / safe + 2.0
    hb = (r - g) / safe + 4.0
    h = torch.where(mx == r, hr, torch.where(mx == g, hg, hb))
    h = torch.where(d == 0, torch.zeros_like(h), h) * 60.0
    s = torch.where(mx == 0, torch.zeros_like(mx), d / torch.where(mx == 0, torch.ones_like(mx), mx))
    return torch.stack([h, s, mx], dim=1)


HED_FROM_RGB = torch.tensor([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11], [0.27, 0.57, 0.78]], dtype=torch.float64).inverse().to(torch.float32)


def hed_from_rgb_spec(p):                       # same definition as oracle orc_hed
    od = (torch.log(p.clamp_min(1e-6).double()) / math.log(1e-6)).to(torch.float32)
    M = HED_FROM_RGB
    out = []
    for k in range(3):
        s = od[:, 0] * M[0, k]
        s = s + od[:, 1] * M[1, k]
        s = s + od[:, 2] * M[2, k]
        out.append(s.clamp_min(0.0))
    return torch.stack(out, dim=1)
