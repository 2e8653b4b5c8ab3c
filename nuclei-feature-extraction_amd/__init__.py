"""nuclei-feature-extraction_amd -- host-side mirror of the reference's feature-set interface
over the MI355X-native engine (libnfx.so, C ABI in include/nfx.h).

The reference's plugin surface (src/features/mod.rs:12-28 of oxabz/nuclei-feature-extraction)::

    trait FeatureSet { fn name(&self) -> &str;
                       fn compute_features_batched(&self, centroids, polygons, patchs, masks) -> DataFrame }

is mirrored by :class:`FeatureSet` and its five implementations with the reference's names
(``ShapeFeatureSet``, ``ColorFeatureSet``, ``GlcmFeatureSet``, ``GLRLMFeatureSet``,
``GaborFilterFeatureSet``); ``feature_sets_from_names`` mirrors ``args::FeatureSet::to_fs``
(args.rs:51-73).  :meth:`Context.extract` is the fused pipeline body (main.rs:146-151).

This module is plumbing only: every number is produced by the HIP kernels behind the C ABI.
There is NO CPU fallback: if libnfx.so is missing or no GPU is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import sys
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NFX_LIB_PATH", os.path.join(_HERE, "libnfx.so"))  # override: kernel experiments only

NFX_MEM_HOST, NFX_MEM_DEVICE = 0, 1
GEOMETRY, COLOR, GLCM, GLRLM, GABOR = range(5)
_STATUS = {0: "NFX_OK", 1: "NFX_ERR_INVALID_ARGUMENT", 2: "NFX_ERR_UNSUPPORTED", 3: "NFX_ERR_DEVICE", 4: "NFX_ERR_OUT_OF_MEMORY"}

# every symbol include/nfx.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "nfx_version", "nfx_build_stamp", "nfx_feature_set_count", "nfx_feature_set_name", "nfx_feature_set_cli_name",
    "nfx_feature_set_num_columns", "nfx_feature_set_column_name", "nfx_parse_feature_sets", "nfx_centroid_key",
    "nfx_ctx_create", "nfx_ctx_destroy", "nfx_last_error", "nfx_ctx_set_stream", "nfx_ctx_synchronize", "nfx_ctx_set_option",
    "nfx_ctx_set_patch_addressing", "nfx_ctx_reserve", "nfx_ctx_workspace_generation", "nfx_feature_set_column_dtype",
    "nfx_polygon_features_f64", "nfx_debug_texture_counts",
    "nfx_ctx_enable_timing", "nfx_ctx_kernel_timing", "nfx_ctx_reset_timing",
    "nfx_preprocess_polygons", "nfx_rasterize_polygons", "nfx_compute_features_batched",
    "nfx_extract", "nfx_extract_tile", "nfx_gather_patches",
    "nfx_extract_tile_streamed_v2", "nfx_host_alloc", "nfx_host_free", "nfx_host_register", "nfx_host_unregister",
    "nfx_comm_unique_id", "nfx_comm_create", "nfx_comm_destroy", "nfx_comm_rank", "nfx_comm_world", "nfx_gather_rows",
)


# nfx_stream_stats after its leading struct_size (include/nfx.h; tests/test_abi.py compiles the header and compares)
STREAM_STATS_FIELDS = ("image_bytes_h2d", "other_bytes_h2d", "rows_bytes_d2h", "blocks_copied", "blocks_total", "block_slots", "chunks",
                       "image_pinned", "split_gather_passes", "rows_direct")


class NfxError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Loads libnfx.so (built by ``__graft_entry__.build()`` / ``make -C csrc``).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NfxError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                       "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i64, u32p, f32p, u8p, i32p = C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)
    L.nfx_feature_set_name.restype = C.c_char_p
    L.nfx_feature_set_cli_name.restype = C.c_char_p
    L.nfx_last_error.restype = C.c_char_p
    L.nfx_last_error.argtypes = [vp]
    L.nfx_feature_set_column_name.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.nfx_parse_feature_sets.argtypes = [C.POINTER(C.c_char_p), C.c_int, i32p, C.c_int]
    L.nfx_centroid_key.argtypes = [C.c_float, C.c_float, C.c_char_p, C.c_size_t]
    L.nfx_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nfx_ctx_destroy.argtypes = [vp]
    L.nfx_ctx_destroy.restype = None
    L.nfx_ctx_set_stream.argtypes = [vp, vp]
    L.nfx_ctx_set_patch_addressing.argtypes = [vp, C.c_int]
    L.nfx_ctx_synchronize.argtypes = [vp]
    L.nfx_ctx_enable_timing.argtypes = [vp, C.c_int]
    L.nfx_ctx_reset_timing.argtypes = [vp]
    L.nfx_ctx_kernel_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(C.c_double)]
    L.nfx_preprocess_polygons.argtypes = [vp, i64, i64, u32p, f32p, f32p, f32p, C.c_int]
    L.nfx_rasterize_polygons.argtypes = [vp, i64, i64, u32p, f32p, C.c_int, f32p, vp, C.c_int]
    L.nfx_compute_features_batched.argtypes = [vp, C.c_int, i64, i64, f32p, u32p, f32p, f32p, f32p, C.c_int, f32p, C.c_int]
    L.nfx_extract.argtypes = [vp, i32p, C.c_int, i64, i64, u32p, f32p, u8p, C.c_int, C.c_int, f32p, f32p, C.c_int]
    L.nfx_extract_tile.argtypes = [vp, i32p, C.c_int, i64, i64, u32p, f32p, u8p, i64, i64, C.c_int, C.c_int, f32p, f32p, C.c_int]
    L.nfx_gather_patches.argtypes = [vp, i64, f32p, u8p, i64, i64, C.c_int, u8p, C.c_int]
    L.nfx_ctx_reserve.argtypes = [vp, i32p, C.c_int, i64, i64, C.c_int, C.c_int]
    L.nfx_ctx_workspace_generation.argtypes = [vp]
    L.nfx_ctx_workspace_generation.restype = C.c_uint64
    L.nfx_feature_set_column_dtype.argtypes = [C.c_int, C.c_int]
    L.nfx_polygon_features_f64.argtypes = [vp, i64, i64, u32p, f32p, vp, C.c_int]
    L.nfx_debug_texture_counts.argtypes = [vp, i64, i64, u32p, f32p, u8p, C.c_int, vp, vp, C.c_int]
    L.nfx_build_stamp.restype = C.c_char_p
    L.nfx_ctx_set_option.argtypes = [vp, C.c_int, i64]
    L.nfx_extract_tile_streamed_v2.argtypes = [vp, i32p, C.c_int, i64, i64, u32p, f32p, u8p, i64, i64, C.c_int, C.c_int, C.c_int, i64, i64, f32p, f32p, vp]
    L.nfx_host_alloc.argtypes = [C.c_size_t]
    L.nfx_host_alloc.restype = vp
    L.nfx_host_free.argtypes = [vp]
    L.nfx_host_free.restype = None
    L.nfx_host_register.argtypes = [vp, C.c_size_t]
    L.nfx_host_unregister.argtypes = [vp]
    L.nfx_host_unregister.restype = None
    L.nfx_comm_unique_id.argtypes = [vp]
    L.nfx_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.nfx_comm_destroy.argtypes = [vp]
    L.nfx_comm_destroy.restype = None
    L.nfx_comm_rank.argtypes = [vp]
    L.nfx_comm_world.argtypes = [vp]
    L.nfx_gather_rows.argtypes = [vp, vp, f32p, C.c_int, C.POINTER(C.c_int64), f32p, C.c_int]
    _lib = L
    return L


def build_stamp() -> str:
    """nfx_build_stamp(): the hash of the sources the loaded libnfx.so was built from."""
    return lib().nfx_build_stamp().decode()


def source_stamp() -> str:
    """The same hash over the sources on disk (csrc/stamp.py, the Makefile's own computation)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_nfx_stamp", os.path.join(_HERE, "csrc", "stamp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_stamp()


def hip_runtimes_mapped() -> List[str]:
    """Every libamdhip64 mapped into this process (real paths, from /proc/self/maps).

    libnfx.so needs `libamdhip64.so.7` (RUNPATH /opt/rocm/lib).  PyTorch's wheel bundles a runtime of its own under the same
    SONAME.  Loaded AFTER `import torch`, libnfx binds the one torch already mapped: ONE runtime in the process (torch's), pinned
    memory and streams are shared.  Loaded BEFORE torch, libnfx maps /opt/rocm's and a later `import torch` maps its own beside it:
    TWO runtimes, and torch is left with "No HIP GPUs are available".  A torch-free process (the C++ command line, a Rust host)
    holds /opt/rocm's alone."""
    libs = set()
    try:
        with open("/proc/self/maps") as fh:
            for ln in fh:
                if "libamdhip64" in ln:
                    libs.add(os.path.realpath(ln.split()[-1]))
    except OSError:
        pass
    return sorted(libs)


# ---- registry (no GPU needed) ---------------------------------------------------------------
def feature_set_name(set_id: int) -> str:
    return lib().nfx_feature_set_name(set_id).decode()


def feature_set_cli_name(set_id: int) -> str:
    return lib().nfx_feature_set_cli_name(set_id).decode()


def num_columns(set_id: int) -> int:
    return lib().nfx_feature_set_num_columns(set_id)


def column_names(set_id: int) -> List[str]:
    buf = C.create_string_buffer(256)
    out = []
    for j in range(num_columns(set_id)):
        if lib().nfx_feature_set_column_name(set_id, j, buf, 256) < 0:
            raise NfxError("column name")
        out.append(buf.value.decode())
    return out


def column_dtypes(set_id: int) -> List[str]:
    """dtype of every column in the reference's frame ("f32" / "f64"): nfx_feature_set_column_dtype."""
    return ["f64" if lib().nfx_feature_set_column_dtype(set_id, j) == 1 else "f32" for j in range(num_columns(set_id))]


def parse_feature_sets(names: Sequence[str]) -> List[int]:
    """args.rs:18-49: case-insensitive names, ``texture``/``all`` expanded, order kept, no de-dup."""
    arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
    out = (C.c_int * (5 * max(1, len(names))))()
    m = lib().nfx_parse_feature_sets(arr, len(names), out, len(out))
    if m < 0:
        raise ValueError(f"{list(names)}: not a valid feature set")   # args.rs:32
    return [out[i] for i in range(m)]


def centroid_key(x: float, y: float) -> str:
    """utils.rs:226-228: ``"{x},{y}"`` with Rust's f32 Display."""
    buf = C.create_string_buffer(128)
    if lib().nfx_centroid_key(x, y, buf, 128) < 0:
        raise NfxError("centroid key")
    return buf.value.decode()


def centroids_to_key_strings(centroids) -> List[str]:
    return [centroid_key(float(c[0]), float(c[1])) for c in np.asarray(centroids, np.float32)]


# ---- buffers: numpy (host) or torch CUDA tensors (device) --------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


def _prep(x, dtype, device_mode: bool, name: str):
    """contiguous array of the right dtype in the right memory space; never copies across spaces."""
    if x is None:
        return None
    if device_mode:
        import torch
        if not _is_torch(x) or not x.is_cuda:
            raise TypeError(f"{name}: expected a CUDA tensor (all buffers of one call share a memory space)")
        ok = {np.float32: (torch.float32,), np.uint32: (torch.int32, getattr(torch, "uint32", torch.int32)),
              np.uint8: (torch.uint8,), np.uint64: (torch.int64, getattr(torch, "uint64", torch.int64))}[dtype]
        if x.dtype not in ok:
            raise TypeError(f"{name}: expected dtype {ok[0]}, got {x.dtype}")
        return x.contiguous()
    if _is_torch(x):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


class Context:
    """One engine context on one GPU (nfx_ctx).  ``stream``: a raw hipStream_t handle (int), e.g.
    ``torch.cuda.current_stream().cuda_stream``; device-memory calls are enqueued on it."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        # One HIP runtime per process (hip_runtimes_mapped): libnfx loaded after `import torch` shares torch's; loaded before it, a
        # later `import torch` maps a second one and torch finds no GPU.  That order is refused here with its cure, not left to
        # surface as "No HIP GPUs are available" somewhere in torch.
        lib()
        mapped = hip_runtimes_mapped()
        if len(mapped) > 1:
            raise NfxError("two HIP runtimes are mapped into this process (" + ", ".join(mapped) + "): libnfx.so was loaded before "
                           "`import torch`, which then brought its own.  Import torch BEFORE the first call into this package (libnfx "
                           "then binds torch's bundled runtime: one runtime, shared streams and pinned memory), or do not import torch "
                           "in this process at all (libnfx then runs on /opt/rocm's)")
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.init()
        h = C.c_void_p()
        st = lib().nfx_ctx_create(device, C.byref(h))
        if st != 0:
            raise NfxError(f"nfx_ctx_create(device={device}) failed: {_STATUS.get(st, st)} (is a GPU visible?)")
        self._h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            lib().nfx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise NfxError(f"{_STATUS.get(st, st)}: {lib().nfx_last_error(self._h).decode()}")

    def set_stream(self, stream: int):
        self._check(lib().nfx_ctx_set_stream(self._h, C.c_void_p(stream)))

    def set_patch_addressing(self, mode: str):
        """"image": load_image_dataset's window (utils.rs:159-191, default); "slide": load_slide_dataset's (utils.rs:97-122)."""
        if mode not in ("image", "slide"):
            raise ValueError("patch addressing is 'image' or 'slide'")
        self._check(lib().nfx_ctx_set_patch_addressing(self._h, 1 if mode == "slide" else 0))

    def set_option(self, option: str, value: int):
        """nfx_ctx_set_option.  "stream_rows_staged": 1 keeps the streamed entry's rows staged on the device and copied home chunk by
        chunk even when `out` is pinned (0, the default: the kernels write pinned rows straight home)."""
        opts = {"stream_rows_staged": 0}
        if option not in opts:
            raise ValueError(f"unknown option {option!r} (known: {sorted(opts)})")
        self._check(lib().nfx_ctx_set_option(self._h, opts[option], int(value)))

    def synchronize(self):
        self._check(lib().nfx_ctx_synchronize(self._h))

    def reserve(self, sets: Sequence[int], n: int, n_vertices: int, ps: int, batch_size: int = 100):
        """nfx_ctx_reserve: workspace and device tables for device-memory calls of up to this shape, built up front; after it
        such calls never allocate or synchronise (and fail instead of growing)."""
        sets_arr = (C.c_int * len(sets))(*sets)
        self._check(lib().nfx_ctx_reserve(self._h, sets_arr, len(sets), n, n_vertices, ps, batch_size))

    def workspace_generation(self) -> int:
        """nfx_ctx_workspace_generation: reallocations of the device-call workspace so far (a captured graph is valid while it is unchanged)."""
        return int(lib().nfx_ctx_workspace_generation(self._h))

    def enable_timing(self, on: bool = True):
        self._check(lib().nfx_ctx_enable_timing(self._h, int(on)))

    def reset_timing(self):
        self._check(lib().nfx_ctx_reset_timing(self._h))

    def kernel_timing(self) -> Dict[str, Dict[str, float]]:
        name, cnt, ms = C.c_char_p(), C.c_int64(), C.c_double()
        n = lib().nfx_ctx_kernel_timing(self._h, -1, None, None, None)
        out = {}
        for k in range(n):
            lib().nfx_ctx_kernel_timing(self._h, k, C.byref(name), C.byref(cnt), C.byref(ms))
            out[name.value.decode()] = {"launches": cnt.value, "total_ms": ms.value}
        return out

    # -- R1 ---------------------------------------------------------------------------------
    def preprocess_polygons(self, ring_offsets, ring_xy):
        dev = _is_torch(ring_xy) and ring_xy.is_cuda
        off = _prep(ring_offsets, np.uint32, dev, "ring_offsets")
        xy = _prep(ring_xy, np.float32, dev, "ring_xy")
        n, nv = len(off) - 1, int(xy.shape[0])
        if dev:
            import torch
            cen = torch.empty((n, 2), dtype=torch.float32, device=xy.device)
            ctr = torch.empty_like(xy)
        else:
            cen, ctr = np.empty((n, 2), np.float32), np.empty_like(xy)
        self._check(lib().nfx_preprocess_polygons(self._h, n, nv, _ptr(off), _ptr(xy), _ptr(cen), _ptr(ctr), int(dev)))
        return cen, ctr

    # -- R2 ---------------------------------------------------------------------------------
    def rasterize_polygons(self, poly_offsets, poly_xy, ps: int, rows: bool = False):
        dev = _is_torch(poly_xy) and poly_xy.is_cuda
        off = _prep(poly_offsets, np.uint32, dev, "poly_offsets")
        xy = _prep(poly_xy, np.float32, dev, "poly_xy")
        n, nv = len(off) - 1, int(xy.shape[0])
        if dev:
            import torch
            masks = torch.empty((n, 1, ps, ps), dtype=torch.float32, device=xy.device)
            r = torch.empty((n, ps) if ps <= 64 else (n, ps, (ps + 63) // 64), dtype=torch.int64, device=xy.device) if rows else None
        else:
            masks = np.empty((n, 1, ps, ps), np.float32)
            r = np.empty((n, ps) if ps <= 64 else (n, ps, (ps + 63) // 64), np.uint64) if rows else None   # ceil(ps / 64) words per row
        self._check(lib().nfx_rasterize_polygons(self._h, n, nv, _ptr(off), _ptr(xy), ps, _ptr(masks), _ptr(r), int(dev)))
        return (masks, r) if rows else masks

    # -- geometric_features::* at their own width (shape.rs:89-97) ------------------------------------
    def polygon_features_f64(self, poly_offsets, poly_xy):
        dev = _is_torch(poly_xy) and poly_xy.is_cuda
        off = _prep(poly_offsets, np.uint32, dev, "poly_offsets")
        xy = _prep(poly_xy, np.float32, dev, "poly_xy")
        n, nv = len(off) - 1, int(xy.shape[0])
        if dev:
            import torch
            out = torch.empty((n, 7), dtype=torch.float64, device=xy.device)
        else:
            out = np.empty((n, 7), np.float64)
        self._check(lib().nfx_polygon_features_f64(self._h, n, nv, _ptr(off), _ptr(xy), _ptr(out), int(dev)))
        return out

    # -- test hook: the integers behind the texture columns ----------------------------------------------
    def debug_texture_counts(self, ring_offsets, ring_xy, patches_u8, ps: int, glcm: bool = True, glrlm: bool = True):
        """(glcm_sums int64 [n,4 levels,4 offsets,6], glrlm_bins uint32 [n,4 directions,24,16]) -- nfx_debug_texture_counts."""
        off = _prep(ring_offsets, np.uint32, False, "ring_offsets")
        xy = _prep(ring_xy, np.float32, False, "ring_xy")
        pu = _prep(patches_u8, np.uint8, False, "patches_u8")
        n, nv = len(off) - 1, int(xy.shape[0])
        g = np.zeros((n, 4, 4, 6), np.int64) if glcm else None
        r = np.zeros((n, 4, 24, 16), np.uint32) if glrlm else None
        self._check(lib().nfx_debug_texture_counts(self._h, n, nv, _ptr(off), _ptr(xy), _ptr(pu), ps, _ptr(g), _ptr(r), 0))
        return g, r

    # -- trait FeatureSet::compute_features_batched ---------------------------------------------
    def compute_features_batched(self, set_id: int, poly_offsets, poly_xy, patchs, masks, centroids=None):
        dev = _is_torch(masks) and masks.is_cuda
        off = _prep(poly_offsets, np.uint32, dev, "poly_offsets")
        xy = _prep(poly_xy, np.float32, dev, "poly_xy")
        p = _prep(patchs, np.float32, dev, "patchs")
        m = _prep(masks, np.float32, dev, "masks")
        # the reference's asserts (shape.rs:23-47, color.rs:18-42)
        assert p is None or p.ndim == 4, "The patchs tensor must be 4 dimensional"
        assert m.ndim == 4, "The masks tensor must be 4 dimensional"
        assert p is None or p.shape[1] == 3, "The patchs tensor must have 3 channels"
        assert m.shape[1] == 1, "The masks tensor must have 1 channel"
        assert p is None or p.shape[0] == m.shape[0], "The number of patchs and masks must be the same"
        n, ps = int(m.shape[0]), int(m.shape[3])
        assert len(off) - 1 == n, "The number of patchs and polygons must be the same"
        F = num_columns(set_id)
        if dev:
            import torch
            out = torch.empty((n, F), dtype=torch.float32, device=m.device)
        else:
            out = np.empty((n, F), np.float32)
        self._check(lib().nfx_compute_features_batched(self._h, set_id, n, int(xy.shape[0]), None, _ptr(off), _ptr(xy),
                                                       _ptr(p), _ptr(m), ps, _ptr(out), int(dev)))
        return out

    # -- pipeline body main.rs:146-151 -------------------------------------------------------
    def extract(self, sets: Sequence[int], ring_offsets, ring_xy, patches_u8, ps: int, batch_size: int = 100,
                out=None, want_centroids: bool = True):
        dev = _is_torch(ring_xy) and ring_xy.is_cuda
        off = _prep(ring_offsets, np.uint32, dev, "ring_offsets")
        xy = _prep(ring_xy, np.float32, dev, "ring_xy")
        pu = _prep(patches_u8, np.uint8, dev, "patches_u8")
        n, nv = len(off) - 1, int(xy.shape[0])
        F = sum(num_columns(s) for s in sets)
        sets_arr = (C.c_int * len(sets))(*sets)
        if dev:
            import torch
            if out is None:
                out = torch.empty((n, F), dtype=torch.float32, device=xy.device)
            cen = torch.empty((n, 2), dtype=torch.float32, device=xy.device) if want_centroids else None
        else:
            if out is None:
                out = np.empty((n, F), np.float32)
            cen = np.empty((n, 2), np.float32) if want_centroids else None
        self._check(lib().nfx_extract(self._h, sets_arr, len(sets), n, nv, _ptr(off), _ptr(xy), _ptr(pu), ps, batch_size,
                                      _ptr(out), _ptr(cen), int(dev)))
        return out, cen

    def extract_tile(self, sets: Sequence[int], ring_offsets, ring_xy, image_u8, ps: int, batch_size: int = 100):
        dev = _is_torch(ring_xy) and ring_xy.is_cuda
        off = _prep(ring_offsets, np.uint32, dev, "ring_offsets")
        xy = _prep(ring_xy, np.float32, dev, "ring_xy")
        img = _prep(image_u8, np.uint8, dev, "image_u8")
        assert img.ndim == 3 and img.shape[0] == 3, "image must be [3,H,W] u8"
        n, nv = len(off) - 1, int(xy.shape[0])
        F = sum(num_columns(s) for s in sets)
        sets_arr = (C.c_int * len(sets))(*sets)
        if dev:
            import torch
            out = torch.empty((n, F), dtype=torch.float32, device=xy.device)
            cen = torch.empty((n, 2), dtype=torch.float32, device=xy.device)
        else:
            out, cen = np.empty((n, F), np.float32), np.empty((n, 2), np.float32)
        self._check(lib().nfx_extract_tile(self._h, sets_arr, len(sets), n, nv, _ptr(off), _ptr(xy), _ptr(img),
                                           int(img.shape[1]), int(img.shape[2]), ps, batch_size, _ptr(out), _ptr(cen), int(dev)))
        return out, cen

    def extract_tile_streamed(self, sets: Sequence[int], ring_offsets, ring_xy, image_u8, ps: int, batch_size: int = 100, block: int = 0,
                              chunk_nuclei: int = 0, cache_bytes: int = 0, out=None):
        """nfx_extract_tile_streamed_v2: host numpy arrays in (image [3,H,W] u8 -- a :func:`pinned_empty` array for the link's full
        rate), rows / centroids / what-travelled out.  Bit-identical to :meth:`extract_tile` on the resident image."""
        off = _prep(ring_offsets, np.uint32, False, "ring_offsets")
        xy = _prep(ring_xy, np.float32, False, "ring_xy")
        if not (isinstance(image_u8, np.ndarray) and image_u8.dtype == np.uint8 and image_u8.flags.c_contiguous):
            image_u8 = _prep(image_u8, np.uint8, False, "image_u8")       # (keeps a pinned array as it is)
        assert image_u8.ndim == 3 and image_u8.shape[0] == 3, "image must be [3,H,W] u8"
        n, nv = len(off) - 1, int(xy.shape[0])
        F = sum(num_columns(s) for s in sets)
        sets_arr = (C.c_int * len(sets))(*sets)
        if out is None:
            out = np.empty((n, F), np.float32)
        assert out.shape == (n, F) and out.dtype == np.float32 and out.flags.c_contiguous
        cen = np.empty((n, 2), np.float32)
        st = (C.c_int64 * (1 + len(STREAM_STATS_FIELDS)))()
        st[0] = C.sizeof(st)                                          # nfx_stream_stats.struct_size: the library writes no further
        self._check(lib().nfx_extract_tile_streamed_v2(self._h, sets_arr, len(sets), n, nv, _ptr(off), _ptr(xy), _ptr(image_u8), int(image_u8.shape[1]),
                                                    int(image_u8.shape[2]), ps, batch_size, block, chunk_nuclei, cache_bytes, _ptr(out), _ptr(cen), st))
        return out, cen, {k: int(st[1 + j]) for j, k in enumerate(STREAM_STATS_FIELDS)}

    def gather_patches(self, centroids, image_u8, ps: int):
        dev = _is_torch(image_u8) and image_u8.is_cuda
        cen = _prep(centroids, np.float32, dev, "centroids")
        img = _prep(image_u8, np.uint8, dev, "image_u8")
        n = int(cen.shape[0])
        if dev:
            import torch
            out = torch.empty((n, 3, ps, ps), dtype=torch.uint8, device=img.device)
        else:
            out = np.empty((n, 3, ps, ps), np.uint8)
        self._check(lib().nfx_gather_patches(self._h, n, _ptr(cen), _ptr(img), int(img.shape[1]), int(img.shape[2]), ps, _ptr(out), int(dev)))
        return out


# ---- pinned host memory (nfx_host_alloc = hipHostMalloc of the runtime libnfx is bound to; in a torch-first process that is torch's own
# runtime, so torch's pinned tensors are pinned for libnfx as well -- hip_runtimes_mapped) ----
class _PinnedBuffer:
    def __init__(self, nbytes):
        self.ptr = lib().nfx_host_alloc(max(1, nbytes))
        if not self.ptr:
            raise MemoryError(f"nfx_host_alloc({nbytes})")
        self.buf = (C.c_uint8 * max(1, nbytes)).from_address(self.ptr)

    def __del__(self):
        try:
            if self.ptr:
                lib().nfx_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """An uninitialised numpy array in memory pinned by nfx_host_alloc (freed with the array's last reference)."""
    # (works without torch: the C++ command line and torch-free hosts pin their slide and rows the same way)
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    owner = _PinnedBuffer(n)
    owner.buf._nfx_owner = owner   # the array keeps owner.buf (its base) alive, owner.buf keeps the owner: the memory outlives every view
    return np.frombuffer(owner.buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


# ---- multi-GPU: the gather of rows over RCCL / xGMI below any framework (nfx_comm_*, nfx_gather_rows) -----------
UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """nfx_comm_unique_id: the 128 bytes one rank makes and every rank passes to :class:`Comm` (distribute them out of band)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    st = lib().nfx_comm_unique_id(buf)
    if st != 0:
        raise NfxError(f"nfx_comm_unique_id: {_STATUS.get(st, st)} (is librccl there?)")
    return buf.raw


class Comm:
    """One rank of a communicator on ``ctx``'s GPU (nfx_comm_create is collective: it returns once every rank has called it)."""

    def __init__(self, ctx: Context, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique_id: 128 bytes from comm_unique_id()")
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        ctx._check(lib().nfx_comm_create(ctx._h, rank, world, C.c_char_p(unique_id), C.byref(h)))
        self._h = h

    def gather_rows(self, rows, counts: Sequence[int], dst=None, root: int = 0):
        """nfx_gather_rows: ``rows`` [counts[rank], F] CUDA f32 tensor; on the root ``dst`` [sum(counts), F] receives every rank's
        rows in rank (= nucleus index) order.  Stream-ordered on the context's stream; returns ``dst``."""
        if len(counts) != self.world:
            raise ValueError("counts: one entry per rank")
        F = int(rows.shape[1])
        if int(rows.shape[0]) != int(counts[self.rank]):
            raise ValueError("rows: counts[rank] rows expected")
        rows = _prep(rows, np.float32, True, "rows")
        if self.rank == root:
            if dst is None:
                import torch
                dst = torch.empty((int(sum(counts)), F), dtype=torch.float32, device=rows.device)
            dst = _prep(dst, np.float32, True, "dst")
        arr = (C.c_int64 * self.world)(*[int(c) for c in counts])
        self.ctx._check(lib().nfx_gather_rows(self.ctx._h, self._h, _ptr(rows), F, arr, _ptr(dst) if dst is not None else None, root))
        return dst

    def close(self):
        if getattr(self, "_h", None):
            lib().nfx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the reference's plugin surface ---------------------------------------------------------------
def _pack_polygons(polygons):
    counts = [len(p) for p in polygons]
    off = np.zeros(len(polygons) + 1, np.uint32)
    np.cumsum(counts, out=off[1:])
    xy = np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in polygons]) if polygons else np.zeros((0, 2), np.float32)
    return off, xy


class FeatureSet:
    """features/mod.rs:12-28.  ``compute_features_batched`` returns the reference's DataFrame as an
    ordered dict: ``"centroid"`` (key strings, utils.rs:226-232) first, then the set's columns."""

    set_id = -1

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def name(self) -> str:
        return feature_set_name(self.set_id)

    def compute_features_batched(self, centroids, polygons, patchs, masks) -> Dict[str, object]:
        n = len(centroids)
        assert n == len(polygons), "The number of patchs and polygons must be the same"       # shape.rs:43-46
        assert int(masks.shape[0]) == n, "The number of patchs and centroids must be the same"  # shape.rs:39-42
        off, xy = _pack_polygons(polygons)
        if _is_torch(masks) and masks.is_cuda:
            import torch
            off_d = torch.from_numpy(off.astype(np.int32)).to(masks.device)
            xy_d = torch.from_numpy(xy).to(masks.device)
            rows = self.ctx.compute_features_batched(self.set_id, off_d, xy_d, patchs, masks)
            self.ctx.synchronize()
            rows = rows.cpu().numpy()
        else:
            rows = self.ctx.compute_features_batched(self.set_id, off, xy, patchs, masks)
        frame: Dict[str, object] = {"centroid": centroids_to_key_strings(centroids)}
        for j, col in enumerate(column_names(self.set_id)):
            frame[col] = rows[:, j].copy()
        return frame


class ShapeFeatureSet(FeatureSet):        # shape.rs:13
    set_id = GEOMETRY


class ColorFeatureSet(FeatureSet):        # color.rs:8
    set_id = COLOR


class GlcmFeatureSet(FeatureSet):         # texture.rs:22
    set_id = GLCM


class GLRLMFeatureSet(FeatureSet):        # texture.rs:178
    set_id = GLRLM


class GaborFilterFeatureSet(FeatureSet):  # texture.rs:317
    set_id = GABOR


_SET_CLASSES = (ShapeFeatureSet, ColorFeatureSet, GlcmFeatureSet, GLRLMFeatureSet, GaborFilterFeatureSet)


def feature_sets_from_names(ctx: Context, names: Sequence[str]) -> List[FeatureSet]:
    """args::FeatureSet::to_fs (args.rs:51-73)."""
    return [_SET_CLASSES[s](ctx) for s in parse_feature_sets(names)]


def all_column_names(sets: Sequence[int]) -> List[str]:
    """Header of the concatenated frame (main.rs:76-89): ``centroid`` then every set's columns."""
    cols = ["centroid"]
    for s in sets:
        cols += column_names(s)
    return cols
