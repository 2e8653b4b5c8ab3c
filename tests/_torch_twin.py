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


# ---- the texture sets as DENSE TENSOR PROGRAMS (the libtorch-shaped CPU baseline; tests/_twin_baseline.py) -------------------------
# texture.rs:25-167,181-310,323-369 is in-tree: gray conversion, the loops over levels / offsets / directions, one `Vec::<f32>::from`
# per feature.  What it calls -- tch_utils::glcm::{glcm, glcm_features}, glrlm::{glrlm, glrlm_features}, gabor::apply_gabor_filter -- is
# not (source absent); these are tensor programs of the SHAPE the call sites fix: a dense `[N,L,L]` f32 co-occurrence tensor per
# (level, offset) built with index_put_(accumulate) and reduced fourteen times, a dense `[N,24,16]` run-length tensor per direction,
# one conv2d to 48 channels.  Same definitions as the oracle (checked in tests/test_torch_twin.py), f32 dense arithmetic as libtorch
# would carry it.  A baseline's workload shape, not a parity reference.
GLCM_LEVELS = (32, 64, 128, 254)                     # texture.rs:19
GLCM_OFFSETS = ((0, 1), (1, 1), (1, 0), (1, -1))     # texture.rs:20 (dy, dx)
GLRLM_LEVELS, GLRLM_MAX_LENGTH = 24, 16              # texture.rs:174-175
GLRLM_DIRECTIONS = ((1, 0), (1, 1), (0, 1), (-1, 1))  # texture.rs:176 (dx, dy)


def quantize(gray, levels):
    return (gray * float(levels)).floor().clamp_(0, levels - 1).to(torch.int64)


def _shifted(t, dy, dx):
    """(t[..., r, c], t[..., r+dy, c+dx]) over the positions where both lie inside the patch"""
    H, W = t.shape[-2], t.shape[-1]
    r0, r1 = max(0, -dy), H - max(0, dy)
    c0, c1 = max(0, -dx), W - max(0, dx)
    return t[..., r0:r1, c0:c1], t[..., r0 + dy:r1 + dy, c0 + dx:c1 + dx]


def glcm_dense(gray, offset, levels, masks, symmetric=True):
    """glcm(&gray, offset, levels, Some(masks), true) -> [N, L, L] f32, normalised (texture.rs:40-46)."""
    n = gray.shape[0]
    q = quantize(gray, levels).squeeze(1)
    a, b = _shifted(q, offset[0], offset[1])
    ma, mb = _shifted(masks.squeeze(1), offset[0], offset[1])
    w = (ma * mb).reshape(-1)
    idx_n = torch.arange(n).view(n, 1, 1).expand_as(a).reshape(-1)
    g = torch.zeros((n, levels, levels), dtype=torch.float32)
    g.index_put_((idx_n, a.reshape(-1), b.reshape(-1)), w, accumulate=True)
    if symmetric:
        g = g + g.transpose(1, 2)
    return g / g.sum(dim=[1, 2], keepdim=True)


def _xlogx(p):
    return torch.where(p > 0, p * p.clamp_min(1e-38).log(), torch.zeros_like(p))


def glcm_features_dense(P):
    """glcm_features(&glcm): fourteen reductions over the dense [N, L, L] tensor, in the column order of texture.rs:81-157."""
    n, L = P.shape[0], P.shape[1]
    i = torch.arange(L, dtype=torch.float32).view(1, L, 1)
    j = torch.arange(L, dtype=torch.float32).view(1, 1, L)
    px = P.sum(dim=2)                                                   # [N, L]
    lv = torch.arange(L, dtype=torch.float32).view(1, L)
    mu = (lv * px).sum(dim=1)
    var = ((lv - mu.view(n, 1)).square() * px).sum(dim=1)
    sij = (i * j * P).sum(dim=[1, 2])
    entropy = -_xlogx(P).sum(dim=[1, 2])
    asm = P.square().sum(dim=[1, 2])
    d = (i - j).abs()
    idm = (P / (1.0 + d.square())).sum(dim=[1, 2])
    flat = P.reshape(n, L * L)
    k_sum = (torch.arange(L).view(L, 1) + torch.arange(L).view(1, L)).reshape(-1)
    k_dif = (torch.arange(L).view(L, 1) - torch.arange(L).view(1, L)).abs().reshape(-1)
    pxy = torch.zeros((n, 2 * L - 1), dtype=torch.float32).index_add_(1, k_sum, flat)
    pd = torch.zeros((n, L), dtype=torch.float32).index_add_(1, k_dif, flat)
    ks = torch.arange(2 * L - 1, dtype=torch.float32).view(1, -1)
    sum_average = (ks * pxy).sum(dim=1)
    sum_variance = ((ks - sum_average.view(n, 1)).square() * pxy).sum(dim=1)
    sum_entropy = -_xlogx(pxy).sum(dim=1)
    contrast = (lv.square() * pd).sum(dim=1)
    dissimilarity = (lv * pd).sum(dim=1)
    difference_variance = ((lv - dissimilarity.view(n, 1)).square() * pd).sum(dim=1)
    hx = -_xlogx(px).sum(dim=1)
    pp = px.view(n, L, 1) * px.view(n, 1, L)
    lpp = torch.where(pp > 0, pp.clamp_min(1e-38).log(), torch.zeros_like(pp))
    hxy1 = -(P * lpp).sum(dim=[1, 2])
    hxy2 = -(pp * lpp).sum(dim=[1, 2])
    correlation = (sij - mu.square()) / var
    imc1 = (entropy - hxy1) / hx
    imc2 = (1.0 - (-2.0 * (hxy2 - entropy)).exp()).clamp_min(0.0).sqrt()
    return [correlation, contrast, dissimilarity, entropy, asm, sum_average, sum_variance, sum_entropy, var, idm,
            dissimilarity, difference_variance, imc1, imc2]


def glcm_set(patchs, masks):                    # texture.rs:25-167 -> [N, 224]
    gs = gray(patchs)
    cols = []
    for levels in GLCM_LEVELS:
        for offset in GLCM_OFFSETS:
            for f in glcm_features_dense(glcm_dense(gs, offset, levels, masks, True)):
                cols.append(torch.tensor(f.tolist(), dtype=torch.float32))    # Vec::<f32>::from(feature): one host copy per column (texture.rs:64-79)
    return torch.stack(cols, dim=1)


def glrlm_dense(gray_scale, levels, max_length, direction, masks):
    """glrlm(&gs, 24, 16, direction, Some(masks)) -> [N, 24, 16] run counts (texture.rs:193-194): a run starts where the predecessor
    along the direction is outside the mask / patch or of another level; its length is found by max_length - 1 shifted compares."""
    n, H, W = gray_scale.shape[0], gray_scale.shape[-2], gray_scale.shape[-1]
    dx, dy = direction
    m = masks.squeeze(1) != 0
    q = torch.where(m, quantize(gray_scale, levels).squeeze(1), torch.full((1,), -1, dtype=torch.int64))
    pad = max_length
    qp = torch.nn.functional.pad(q, (pad, pad, pad, pad), value=-2)     # a border no level equals
    core = lambda sy, sx: qp[:, pad + sy:pad + sy + H, pad + sx:pad + sx + W]
    start = m & (core(-dy, -dx) != q)
    alive = start.clone()
    length = start.to(torch.int64)
    for s in range(1, max_length):
        alive = alive & (core(s * dy, s * dx) == q)
        length += alive.to(torch.int64)
    idx_n = torch.arange(n).view(n, 1, 1).expand_as(q)
    sel = start.reshape(-1)
    r = torch.zeros((n, levels, max_length), dtype=torch.int64)
    r.index_put_((idx_n.reshape(-1)[sel], q.reshape(-1)[sel], (length.reshape(-1)[sel] - 1)), torch.ones(int(sel.sum()), dtype=torch.int64), accumulate=True)
    return r


def glrlm_features_dense(R, pixel_count):
    """glrlm_features(&glrlm, Some(&pixel_count)) on the f32 [N, 24, 16] tensor: 17 columns in the order of texture.rs:244-301."""
    n, L, M = R.shape
    g = torch.arange(1, L + 1, dtype=torch.float32).view(1, L, 1)
    l = torch.arange(1, M + 1, dtype=torch.float32).view(1, 1, M)
    gg, ll = g.square(), l.square()
    e = (g - (L + 1) / 2.0).square()
    wmid, wext = 1.0 / (1.0 + e), e
    nr = R.sum(dim=[1, 2])
    s = lambda t: t.sum(dim=[1, 2]) / nr
    rlm = s(R * l)
    return [s(R / ll), s(R * ll), R.sum(dim=2).square().sum(dim=1) / nr, R.sum(dim=1).square().sum(dim=1) / nr, s(R / gg), s(R * gg),
            s(R / (ll * gg)), s(R * gg / ll), s(R * ll / gg), s(R * ll * gg), s(R * wmid / ll), s(R * wmid * ll), s(R * wext / ll), s(R * wext * ll),
            nr / pixel_count, rlm, s(R * (l - rlm.view(n, 1, 1)).square())]


def glrlm_set(patchs, masks):                   # texture.rs:181-310 -> [N, 68]
    gs = gray(patchs)
    pixel_count = masks.sum(dim=[-3, -2, -1])
    cols = []
    for direction in GLRLM_DIRECTIONS:
        R = glrlm_dense(gs, GLRLM_LEVELS, GLRLM_MAX_LENGTH, direction, masks).to(torch.float32)
        for f in glrlm_features_dense(R, pixel_count):
            cols.append(torch.tensor(f.tolist(), dtype=torch.float32))
    return torch.stack(cols, dim=1)


def gabor_set(patchs, masks, bank):             # texture.rs:323-369 -> [N, 96]; bank [48,1,30,30] = apply_gabor_filter's kernels
    gs = gray(patchs)
    filtered = torch.nn.functional.conv2d(torch.nn.functional.pad(gs, (15, 14, 15, 14)), bank)
    return gabor_stats(filtered, masks)
