"""Synthetic nuclei (GeoJSON-shaped outer rings) and H&E-like patches / tiles.

Follows SURVEY.md §8d: counter-based RNG (splitmix64 of ``seed ^ counter``) so any index shard
regenerates exactly its slice; star-convex polygons with ``V ~ U{12..48}`` vertices plus the
closing duplicate (the reference sums *all* listed ring vertices, utils.rs:54-64 of the
reference); radius ``rho(t) = rho0 * (1 + 0.25 * sum_{k=2..4} a_k cos(k t + phi_k))`` with
``rho0 ~ U(0.15, 0.40) * ps``; coordinates rounded to f32.

Everything is written against int64 two's-complement arithmetic so the same code runs on numpy
(CPU tests, oracle inputs) and on torch tensors (bench: data generated directly in HBM).
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5EED0000
_GOLD = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687            # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685            # 0x94D049BB133111EB


def _lsr(x, s):
    """logical shift right on int64 two's complement (numpy or torch)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x):
    """splitmix64 finaliser on int64 arrays/tensors (wrapping arithmetic)."""
    z = x + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _u01(z, xp):
    """53-bit uniform in [0,1) from a mixed int64."""
    v = _lsr(z, 11)
    if xp is np:
        return v.astype(np.float64) * (1.0 / 9007199254740992.0)
    return v.to(xp.float64) * (1.0 / 9007199254740992.0)


def _ctr(seed, idx, stream, xp):
    """counter for (nucleus idx, stream id)."""
    with np.errstate(over="ignore"):
        return mix64((idx * 1024 + stream) ^ seed)


def _arange(i0, n, xp, device=None):
    if xp is np:
        return np.arange(i0, i0 + n, dtype=np.int64)
    return xp.arange(i0, i0 + n, dtype=xp.int64, device=device)


MAX_V = 49  # 48 + closing duplicate


def nucleus_params(seed, i0, n, ps, xp=np, device=None):
    """Per-nucleus scalars: V (without closing point), rho0, a[3], phi[3]."""
    idx = _arange(i0, n, xp, device)
    with np.errstate(over="ignore"):
        u = [_u01(_ctr(seed, idx, s, xp), xp) for s in range(8)]
    V = 12 + (u[0] * 37.0)
    V = V.astype(np.int64) if xp is np else V.to(xp.int64)   # 12..48
    rho0 = (0.15 + 0.25 * u[1]) * ps
    a = [u[2], u[3], u[4]]
    phi = [u[5] * (2 * np.pi), u[6] * (2 * np.pi), u[7] * (2 * np.pi)]
    return idx, V, rho0, a, phi


def _radius(theta, rho0, a, phi, xp):
    s = 0.0
    for k, (ak, pk) in zip((2, 3, 4), zip(a, phi)):
        s = s + ak * xp.cos(k * theta + pk)
    return rho0 * (1.0 + 0.25 * s)


def centres(seed, i0, n, ps, tile_w, tile_h):
    """Nucleus centres (x, y) in slide pixels, f64.  98 % keep a ps/2 margin from the tile border,
    2 % fall inside the margin to exercise the zero-padding rules (utils.rs:174-191)."""
    idx = _arange(i0, n, np)
    with np.errstate(over="ignore"):
        ux = _u01(_ctr(seed, idx, 8, np), np)
        uy = _u01(_ctr(seed, idx, 9, np), np)
        um = _u01(_ctr(seed, idx, 10, np), np)
    margin = np.where(um < 0.02, 0.0, ps / 2.0)
    cx = margin + ux * (tile_w - 2 * margin)
    cy = margin + uy * (tile_h - 2 * margin)
    return cx, cy


def polygons(seed, i0, n, ps, cx=None, cy=None):
    """Raw GeoJSON-style outer rings for nuclei [i0, i0+n).

    Returns (ring_offsets uint32 [n+1], ring_xy float32 [sum V, 2]).  The closing duplicate of the
    first vertex is included, as QuPath/GeoJSON rings have it.
    """
    idx, V, rho0, a, phi = nucleus_params(seed, i0, n, ps)
    if cx is None:
        cx = np.full(n, 1000.0) + (idx % 7) * 0.37
        cy = np.full(n, 1000.0) + (idx % 5) * 0.61
    counts = V + 1
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(counts, out=off[1:])
    total = int(off[-1])
    owner = np.repeat(np.arange(n), counts)
    k = np.arange(total) - off[:-1][owner].astype(np.int64)
    k = np.where(k == V[owner], 0, k)                         # closing point = vertex 0
    theta = 2 * np.pi * k / V[owner]
    rad = _radius(theta, rho0[owner], [x[owner] for x in a], [x[owner] for x in phi], np)
    xy = np.empty((total, 2), dtype=np.float32)
    xy[:, 0] = (cx[owner] + rad * np.cos(theta)).astype(np.float32)
    xy[:, 1] = (cy[owner] + rad * np.sin(theta)).astype(np.float32)
    return off, xy


def patches(seed, i0, n, ps, xp=np, device=None, chunk=None):
    """H&E-like packed patches u8 [n,3,ps,ps] (planar, the reference's tensor layout utils.rs:198).

    Background (235,220,235) +-8; nucleus interior (the analytic star shape of the same nucleus,
    centred in the patch) darker/bluer with per-nucleus chroma and a two-octave value-noise
    texture, so GLCM/GLRLM/Gabor statistics are non-degenerate.
    """
    if chunk is None:
        chunk = 4096 if xp is np else 65536
    outs = []
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        outs.append(_patch_chunk(seed, i0 + c0, m, ps, xp, device))
    if not outs:
        if xp is np:
            return np.zeros((0, 3, ps, ps), np.uint8)
        return xp.zeros((0, 3, ps, ps), dtype=xp.uint8, device=device)
    return np.concatenate(outs) if xp is np else xp.cat(outs)


def _patch_chunk(seed, i0, n, ps, xp, device):
    idx, V, rho0, a, phi = nucleus_params(seed, i0, n, ps, xp, device)
    rr = _arange(0, ps, xp, device)
    f64 = np.float64 if xp is np else xp.float64

    def f(x):
        return x.astype(f64) if xp is np else x.to(f64)

    y = (f(rr) - ps / 2.0).reshape(1, ps, 1)
    x = (f(rr) - ps / 2.0).reshape(1, 1, ps)
    theta = xp.arctan2(y + 0 * x, x + 0 * y) if xp is np else xp.atan2(y + 0 * x, x + 0 * y)
    dist = xp.sqrt(x * x + y * y)
    sh = (n, 1, 1)
    rad = _radius(theta, rho0.reshape(sh), [v.reshape(sh) for v in a], [v.reshape(sh) for v in phi], xp)
    inside = dist <= rad                                            # [n,ps,ps]

    i3 = idx.reshape(sh)
    r3 = rr.reshape(1, ps, 1)
    c3 = rr.reshape(1, 1, ps)
    with np.errstate(over="ignore"):
        base = mix64(i3 * 4099 + (seed ^ 0x51F15E))
        fine = mix64(base + r3 * 131 + c3)
        coarse = mix64(base + (r3 >> 2) * 977 + (c3 >> 2) + 77777)
        chroma = [_u01(_ctr(seed, idx, 16 + ch, xp), xp).reshape(sh) for ch in range(3)]
    ufine = [_u01(mix64(fine + ch), xp) for ch in range(3)]
    ucoarse = _u01(coarse, xp)
    bg = (235.0, 220.0, 235.0)
    lo = (70.0, 40.0, 120.0)
    planes = []
    for ch in range(3):
        back = bg[ch] + (ufine[ch] * 16.0 - 8.0)
        nuc = lo[ch] + 60.0 * chroma[ch] + 40.0 * (ucoarse - 0.5) + (ufine[ch] * 12.0 - 6.0)
        v = xp.where(inside, nuc, back)
        v = xp.clip(v, 0.0, 255.0) if xp is np else v.clamp(0.0, 255.0)
        planes.append(v)
    if xp is np:
        return np.stack(planes, axis=1).astype(np.uint8)
    return xp.stack(planes, dim=1).to(xp.uint8)


def tile(seed, H, W, ring_offsets, ring_xy, ps):
    """RGB u8 tile [3,H,W] (planar, as tch::vision::image::load gives, main.rs:26 of the reference)
    with background noise and darker blobs at the given rings' vertex-mean centres.  CPU / small
    sizes only (config 1 and the patch-addressing tests)."""
    yy = np.arange(H, dtype=np.int64).reshape(H, 1)
    xx = np.arange(W, dtype=np.int64).reshape(1, W)
    with np.errstate(over="ignore"):
        h = mix64(yy * 1000003 + xx + (seed ^ 0x7117E))
    img = np.empty((3, H, W), dtype=np.float64)
    bg = (235.0, 220.0, 235.0)
    for ch in range(3):
        with np.errstate(over="ignore"):
            img[ch] = bg[ch] + _u01(mix64(h + ch), np) * 16.0 - 8.0
    n = len(ring_offsets) - 1
    lo = (70.0, 40.0, 120.0)
    for i in range(n):
        pts = ring_xy[ring_offsets[i]:ring_offsets[i + 1]].astype(np.float64)
        cx, cy = pts[:, 0].mean(), pts[:, 1].mean()
        rad = np.sqrt(((pts - [cx, cy]) ** 2).sum(1)).mean()
        y0, y1 = max(0, int(cy - rad - 1)), min(H, int(cy + rad + 2))
        x0, x1 = max(0, int(cx - rad - 1)), min(W, int(cx + rad + 2))
        if y0 >= y1 or x0 >= x1:
            continue
        sy = np.arange(y0, y1).reshape(-1, 1)
        sx = np.arange(x0, x1).reshape(1, -1)
        blob = ((sy - cy) ** 2 + (sx - cx) ** 2) <= rad * rad
        with np.errstate(over="ignore"):
            tex = _u01(mix64((sy >> 2) * 977 + (sx >> 2) + i * 4099 + seed), np)
        for ch in range(3):
            sub = img[ch, y0:y1, x0:x1]
            sub[blob] = (lo[ch] + 30.0 * ((i * 37 + ch * 11) % 17) / 17.0 + 40.0 * (tex - 0.5))[blob] + (sub[blob] - bg[ch])
    return np.clip(img, 0, 255).astype(np.uint8)


# ---- slide-scale tiles for the streamed workloads (BASELINE configs[4]; bench only) ---------------------------------------
def tile_side(n, ps):
    """logical tile T x T with T = ceil(sqrt(n (ps/1.5)^2)) -- (ps/1.5)^2 slide pixels per nucleus."""
    return int(np.ceil(np.sqrt(n * (ps / 1.5) ** 2)))


def centres_slide_order(seed, i0, n, n_total, ps, tile_w, tile_h, block=2048, slide_i0=0):
    """Centres of nuclei [i0, i0+n) of a slide that holds nuclei [slide_i0, slide_i0 + n_total) listed TILE BY TILE, as
    segmentation tools emit them: the slide's k-th nucleus lies in block floor(k * n_blocks / n_total) of a raster of
    `block`-pixel squares, uniformly inside it (always at least ps/2 + 1 pixels from the slide's border: the border rules are the
    tests' business, not the bench's)."""
    idx = _arange(i0, n, np)
    nbx, nby = -(-tile_w // block), -(-tile_h // block)
    b = ((idx - slide_i0) * (nbx * nby)) // max(1, n_total)
    by, bx = b // nbx, b % nbx
    with np.errstate(over="ignore"):
        ux = _u01(_ctr(seed, idx, 8, np), np)
        uy = _u01(_ctr(seed, idx, 9, np), np)
    m = ps / 2.0 + 1.0
    x0, x1 = np.maximum(bx * block, m), np.minimum((bx + 1) * block, tile_w - m)
    y0, y1 = np.maximum(by * block, m), np.minimum((by + 1) * block, tile_h - m)
    return x0 + ux * np.maximum(x1 - x0, 0.0), y0 + uy * np.maximum(y1 - y0, 0.0)


def tile_from_patches(torch, d_patches, ring_cx, ring_cy, H, W, ps, seed, device, chunk=16384):
    """RGB u8 tile [3,H,W] on `device`: background noise, every nucleus's packed patch pasted where load_image_dataset will
    look for it (top-left = trunc(c - ps/2), utils.rs:159-162).  Windows overlap at slide density; later writes win (which
    one is unspecified -- the tile is made once and every path under test reads the same bytes).  ring_cx / ring_cy: the
    rings' vertex-mean centroids as the engine computes them (f32)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) & 0x7FFFFFFF)
    tile = torch.empty((3, H, W), dtype=torch.uint8, device=device)
    bg = (235, 220, 235)
    for ch in range(3):
        for r0 in range(0, H, 4096):   # in bands: randint's int64 temporaries stay small
            r1 = min(H, r0 + 4096)
            tile[ch, r0:r1] = (torch.randint(0, 17, (r1 - r0, W), generator=g, device=device, dtype=torch.int16) + (bg[ch] - 8)).to(torch.uint8)
    n = int(d_patches.shape[0])
    top = torch.from_numpy(np.trunc(np.asarray(ring_cy, np.float32) - np.float32(ps / 2.0)).astype(np.int64)).to(device)
    left = torch.from_numpy(np.trunc(np.asarray(ring_cx, np.float32) - np.float32(ps / 2.0)).astype(np.int64)).to(device)
    rr = torch.arange(ps, device=device, dtype=torch.int64)
    flat = tile.view(3, H * W)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        y = top[c0:c1].view(-1, 1, 1) + rr.view(1, ps, 1)
        x = left[c0:c1].view(-1, 1, 1) + rr.view(1, 1, ps)
        ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
        pos = (y.clamp(0, H - 1) * W + x.clamp(0, W - 1))[ok]
        for ch in range(3):
            flat[ch].index_put_((pos,), d_patches[c0:c1, ch][ok])
    return tile
