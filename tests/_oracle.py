"""ctypes loader for the CPU oracle (oracle/nfx_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libnfx_oracle.so")
_lib = None

NCOLS = (12, 18, 224, 68, 96)


def build(force: bool = False) -> str:
    src = os.path.join(_ROOT, "oracle", "nfx_oracle.c")
    if force or not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def preprocess_polygon(ring_xy):
    ring_xy = np.ascontiguousarray(ring_xy, np.float32)
    V = ring_xy.shape[0]
    cen = np.zeros(2, np.float32)
    out = np.zeros((V, 2), np.float32)
    lib().orc_preprocess_polygon(_p(ring_xy, C.c_float), C.c_int(V), _p(cen, C.c_float), _p(out, C.c_float))
    return cen, out


def raster_polygon(centred_xy, ps):
    centred_xy = np.ascontiguousarray(centred_xy, np.float32)
    m = np.zeros((ps, ps), np.float32)
    lib().orc_raster_polygon(_p(centred_xy, C.c_float), C.c_int(centred_xy.shape[0]), C.c_int(ps), _p(m, C.c_float))
    return m


def raster_ellipse(ps, xc, yc, a, b, angle):
    m = np.zeros((ps, ps), np.float32)
    lib().orc_raster_ellipse(C.c_int(ps), C.c_float(xc), C.c_float(yc), C.c_float(a), C.c_float(b), C.c_float(angle), _p(m, C.c_float))
    return m


def shape_one(centred_xy, mask):
    centred_xy = np.ascontiguousarray(centred_xy, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    ps = mask.shape[-1]
    out = np.zeros(12, np.float32)
    aux = np.zeros(8, np.float64)
    lib().orc_shape_one(_p(centred_xy, C.c_float), C.c_int(centred_xy.shape[0]), _p(mask, C.c_float), C.c_int(ps),
                        _p(out, C.c_float), _p(aux, C.c_double))
    return out, aux


def polygon_features_f64(centred_xy):
    """geometric_features::* on one centred polygon (shape.rs:89-97): 7 doubles, order as nfx_polygon_features_f64."""
    centred_xy = np.ascontiguousarray(centred_xy, np.float32)
    out = np.zeros(7, np.float64)
    lib().orc_polygon_features_f64(_p(centred_xy, C.c_float), C.c_int(centred_xy.shape[0]), _p(out, C.c_double))
    return out


def compute_features_batched(set_id, poly_offsets, poly_xy, patches, masks):
    """FeatureSet::compute_features_batched for ONE reference batch (features/mod.rs:21-27)."""
    poly_offsets = np.ascontiguousarray(poly_offsets, np.uint32)
    poly_xy = np.ascontiguousarray(poly_xy, np.float32)
    patches = np.ascontiguousarray(patches, np.float32)
    masks = np.ascontiguousarray(masks, np.float32)
    n = patches.shape[0]
    ps = patches.shape[-1]
    out = np.zeros((n, NCOLS[set_id]), np.float32)
    rc = lib().orc_compute_features_batched(C.c_int(set_id), C.c_int(n), _p(poly_offsets, C.c_uint32), _p(poly_xy, C.c_float),
                                            _p(patches, C.c_float), _p(masks, C.c_float), C.c_int(ps), _p(out, C.c_float))
    assert rc == 0
    return out


def extract(sets, ring_offsets, ring_xy, patches_u8, ps, batch_size=100, threads=1):
    """main.rs:146-151 pipeline body on packed u8 patches.  Returns (rows [n,F], centroids [n,2])."""
    sets = np.ascontiguousarray(sets, np.int32)
    ring_offsets = np.ascontiguousarray(ring_offsets, np.uint32)
    ring_xy = np.ascontiguousarray(ring_xy, np.float32)
    patches_u8 = np.ascontiguousarray(patches_u8, np.uint8)
    n = len(ring_offsets) - 1
    F = int(sum(NCOLS[s] for s in sets))
    rows = np.zeros((n, F), np.float32)
    cen = np.zeros((n, 2), np.float32)
    f = lib().orc_extract
    f.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int64, C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                  C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    rc = f(_p(sets, C.c_int), len(sets), n, _p(ring_offsets, C.c_uint32), _p(ring_xy, C.c_float),
           _p(patches_u8, C.c_uint8), ps, batch_size, _p(rows, C.c_float), _p(cen, C.c_float), threads)
    assert rc == 0
    return rows, cen


def glcm_counts(patch, mask, L, dy, dx):
    patch = np.ascontiguousarray(patch, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    ps = mask.shape[-1]
    G = np.zeros((L, L), np.int64)
    lib().orc_glcm_counts(_p(patch, C.c_float), _p(mask, C.c_float), C.c_int(ps), C.c_int(L), C.c_int(dy), C.c_int(dx), _p(G, C.c_int64))
    return G


def glrlm_counts(patch, mask, dx, dy):
    patch = np.ascontiguousarray(patch, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    ps = mask.shape[-1]
    R = np.zeros((24, 16), np.int64)
    lib().orc_glrlm_counts(_p(patch, C.c_float), _p(mask, C.c_float), C.c_int(ps), C.c_int(dx), C.c_int(dy), _p(R, C.c_int64))
    return R


def gabor_bank():
    w = np.zeros((48, 30, 30), np.float32)
    lib().orc_gabor_bank(_p(w, C.c_float))
    return w


def crop_patch(image, cx, cy, ps, slide=False):
    image = np.ascontiguousarray(image, np.uint8)
    _, H, W = image.shape
    out = np.zeros((3, ps, ps), np.uint8)
    (lib().orc_crop_patch_slide if slide else lib().orc_crop_patch)(_p(image, C.c_uint8), C.c_int(H), C.c_int(W), C.c_float(cx), C.c_float(cy), C.c_int(ps), _p(out, C.c_uint8))
    return out


GLCM_LEVELS = (32, 64, 128, 254)                  # texture.rs:19
GLCM_OFFSETS = ((0, 1), (1, 1), (1, 0), (1, -1))  # texture.rs:20 (dy, dx)
GLRLM_DIRS = ((1, 0), (1, 1), (0, 1), (-1, 1))    # texture.rs:176 (dx, dy)


def glcm_invariants(G):
    """The six exact integers nfx_debug_texture_counts reports, from a dense symmetric count matrix G (int64 [L,L])."""
    L = G.shape[0]
    a = np.arange(L, dtype=np.int64)[:, None]
    b = np.arange(L, dtype=np.int64)[None, :]
    s, d = a + b, np.abs(a - b)
    tot = int(G.sum())
    assert tot % 2 == 0
    return np.array([tot // 2, int((s * G).sum()) // 2, int((s * s * G).sum()) // 2, int((d * G).sum()) // 2,
                     int((d * d * G).sum()) // 2, int((G * G).sum())], np.int64)


def texture_counts(ring_offsets, ring_xy, patches_u8, ps):
    """Count-level oracle for the texture sets on packed u8 patches: masks as the pipeline makes them (R1 + R2), then
    (glcm_sums int64 [n,4,4,6], glrlm_bins uint32 [n,4,24,16]) in the layout of nfx_debug_texture_counts."""
    ring_offsets = np.asarray(ring_offsets, np.int64)
    n = len(ring_offsets) - 1
    g = np.zeros((n, 4, 4, 6), np.int64)
    r = np.zeros((n, 4, 24, 16), np.uint32)
    for i in range(n):
        _, ctr = preprocess_polygon(ring_xy[ring_offsets[i]:ring_offsets[i + 1]])
        mask = raster_polygon(ctr, ps)
        patch = patches_u8[i].astype(np.float32) / np.float32(255.0)      # utils.rs:172
        for li, L in enumerate(GLCM_LEVELS):
            for oi, (dy, dx) in enumerate(GLCM_OFFSETS):
                g[i, li, oi] = glcm_invariants(glcm_counts(patch, mask, L, dy, dx))
        for di, (dx, dy) in enumerate(GLRLM_DIRS):
            r[i, di] = glrlm_counts(patch, mask, dx, dy).astype(np.uint32)
    return g, r
