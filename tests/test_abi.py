"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/nfx.h
declares, and the registry reproduces the reference's names, column order and CLI expansion.
No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(nfx):
    hdr = open(os.path.join(ROOT, "include", "nfx.h")).read()
    declared = set(re.findall(r"\b(nfx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"nfx_ctx"}
    assert declared == set(nfx.ABI_SYMBOLS), declared ^ set(nfx.ABI_SYMBOLS)
    L = nfx.lib()
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.nfx_version() == 500
    # the streamed entry of ABI <= 0.4.0 stays exported under its old name (it returns NFX_ERR_UNSUPPORTED: an old binary gets an error,
    # not a write past its shorter nfx_stream_stats)
    old = L.nfx_extract_tile_streamed
    old.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_int64] * 2 + [C.c_void_p] * 3 + [C.c_int64] * 2 + [C.c_int] * 3 + [C.c_int64] * 2 + [C.c_void_p] * 3
    assert old(None, None, 0, 0, 0, None, None, None, 0, 0, 0, 0, 0, 0, 0, None, None, None) == 2
    assert L.nfx_ctx_set_option(None, 0, 0) == 1


def test_the_binary_is_bound_to_the_sources_it_was_built_from(nfx, tmp_path):
    """nfx_build_stamp() -- compiled in by csrc/Makefile -- equals the hash of the sources on disk (csrc/stamp.py): a stale prebuilt
    libnfx.so cannot pass for a build of HEAD's sources.  The same comparison guards bench.py (it refuses to print a line on a
    mismatch): shown here on a copy of the sources with one byte changed."""
    import shutil
    import sys
    assert nfx.build_stamp() == nfx.source_stamp(), "libnfx.so is stale: run `make -C nuclei-feature-extraction_amd/csrc`"
    assert re.fullmatch(r"[0-9a-f]{16}", nfx.build_stamp())
    sys.path.insert(0, ROOT)
    import bench
    assert bench.library_stamp_on_disk() == nfx.build_stamp() and bench.library_is_fresh()      # read out of the file, without loading it
    src = os.path.join(ROOT, "nuclei-feature-extraction_amd", "csrc")
    dst = tmp_path / "pkg" / "csrc"
    dst.parent.mkdir()
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("_obj", "gpurun_out", "__pycache__"))
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "include", "nfx.h"), tmp_path / "include" / "nfx.h")
    sys.path.insert(0, str(dst))
    try:
        import importlib
        stamp = importlib.import_module("stamp")
        assert stamp.source_stamp(str(dst)) == nfx.build_stamp()          # an identical copy hashes the same
        with open(dst / "k_color.hip", "ab") as fh:
            fh.write(b"\n// edited after the build\n")
        stale = stamp.source_stamp(str(dst))
        assert stale != nfx.build_stamp()
    finally:
        sys.path.remove(str(dst))
        sys.modules.pop("stamp", None)
    bench.require_fresh_library(nfx)                                       # this tree: fine
    with pytest.raises(SystemExit) as e:
        bench.require_fresh_library(nfx, source_stamp=stale)               # the edited sources against this binary: refused
    assert "stale" in str(e.value)


def test_the_library_reads_no_environment_but_the_documented_name(nfx):
    """The default build is steered through the C ABI only (nfx_ctx_set_option): no tuning or measurement switch can reach a product
    process through its environment.  Every NFX_* string in the binary is one the header documents."""
    import subprocess
    out = subprocess.check_output(["strings", "-a", nfx.LIB_PATH]).decode(errors="replace")
    names = set(re.findall(r"NFX_[A-Z0-9_]+", out))
    hdr = open(os.path.join(ROOT, "include", "nfx.h")).read()
    assert "NFX_RCCL_PATH" in names
    env_like = {n for n in names if n not in ("NFX_OPT_STREAM_ROWS_STAGED", "NFX_BUILD_STAMP")}   # (an option's name inside an error text; the stamp's marker)
    assert env_like == {"NFX_RCCL_PATH"}, env_like
    assert all(n in hdr for n in names)
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", nfx.LIB_PATH]).decode()
    if "getenv" in syms:   # one call site: nfx_comm.cpp, NFX_RCCL_PATH
        src = os.path.join(ROOT, "nuclei-feature-extraction_amd", "csrc")
        hits = [(fn, ln.strip()) for fn in sorted(os.listdir(src)) if fn.endswith((".hip", ".cpp", ".h"))
                for ln in open(os.path.join(src, fn)) if "getenv(" in ln]
        outside = []
        for fn in sorted({h[0] for h in hits}):
            guarded, depth = False, 0
            for ln in open(os.path.join(src, fn)):
                t = ln.strip()
                if t.startswith("#ifdef NFX_EXPERIMENTS"):
                    guarded = True
                elif guarded and t.startswith(("#else", "#endif")):
                    guarded = False
                if "getenv(" in ln and not guarded:
                    outside.append((fn, t))
        assert [o for o in outside if "NFX_RCCL_PATH" not in o[1]] == [], outside


def test_stream_stats_mirror_matches_the_header(nfx, tmp_path):
    """nfx_stream_stats as the C compiler lays it out == the Python mirror (struct_size first, then STREAM_STATS_FIELDS in order):
    a field added to one and not the other is caught here, not as memory written past a caller's struct."""
    import subprocess
    fields = ("struct_size",) + tuple(nfx.STREAM_STATS_FIELDS)
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nfx.h"\nint main(void){printf("%zu", sizeof(nfx_stream_stats));'
                   + "".join(f'printf(" %zu", offsetof(nfx_stream_stats, {f}));' for f in fields) + 'printf(" %d", NFX_VERSION);return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[0] == 8 * len(fields)
    assert got[1:-1] == [8 * j for j in range(len(fields))]
    assert got[-1] == nfx.lib().nfx_version()


def test_registry_names_and_counts(nfx):
    assert nfx.lib().nfx_feature_set_count() == 5
    # FeatureSet::name(): shape.rs:132-134, color.rs:104-106, texture.rs:169-171,312-314,371-373
    assert [nfx.feature_set_name(s) for s in range(5)] == ["geometry", "color", "GLCM", "GLRLM", "gabor filter"]
    assert [nfx.num_columns(s) for s in range(5)] == [12, 18, 224, 68, 96]
    assert sum(nfx.num_columns(s) for s in range(5)) == 418
    assert nfx.lib().nfx_feature_set_num_columns(7) == -1


def test_column_names_follow_the_reference(nfx):
    g = nfx.column_names(0)
    assert g == ["area", "major_axis", "minor_axis", "eccentricity", "orientation", "perimeter", "equivalent_perimeter",
                 "compacity", "eliptic_deviation", "convex_hull_area", "convex_deffect", "convex_perimeter"]   # shape.rs:113-128
    c = nfx.column_names(1)
    assert c[0] == "mean_r" and c[6] == "mean_h" and c[12] == "mean_haematoxylin" and c[17] == "std_dab"       # color.rs:80-99
    t = nfx.column_names(2)
    assert t[0] == "correlation_0_1_32" and t[13] == "information_measure_correlation2_0_1_32"                  # texture.rs:81-157
    assert t[14] == "correlation_1_1_32" and t[3 * 14] == "correlation_1_-1_32" and t[-1] == "information_measure_correlation2_1_-1_254"
    r = nfx.column_names(3)
    assert r[0] == "short_run_emphasis_1_0" and r[16] == "run_length_variance_1_0" and r[-1] == "run_length_variance_-1_1"  # texture.rs:244-301
    b = nfx.column_names(4)
    assert b[0] == "gabor_angle_0_frequency_0.5_mean" and b[1] == "gabor_angle_0_frequency_0.5_variance"       # texture.rs:346-361
    assert b[2] == "gabor_angle_0_frequency_1_mean" and b[12] == "gabor_angle_45_frequency_0.5_mean" and b[-1] == "gabor_angle_315_frequency_8_variance"
    assert len(set(g + c + t + r + b)) == 418
    assert nfx.all_column_names([0, 1])[0] == "centroid"


def test_feature_set_cli_parsing(nfx):
    # args.rs:18-49
    assert nfx.parse_feature_sets(["geometry"]) == [0]
    assert nfx.parse_feature_sets(["ALL"]) == [0, 1, 2, 3, 4]
    assert nfx.parse_feature_sets(["Texture"]) == [2, 3, 4]
    assert nfx.parse_feature_sets(["color", "all", "glcm"]) == [1, 0, 1, 2, 3, 4, 2]   # no de-dup, order kept
    with pytest.raises(ValueError):
        nfx.parse_feature_sets(["shape"])


def test_centroid_keys_use_rust_f32_display(nfx):
    # utils.rs:226-228, Rust Display for f32: shortest round-trip, positional, no trailing ".0"
    assert nfx.centroid_key(100.0, 2047.96875) == "100,2047.9688"
    assert nfx.centroid_key(0.1, -3.5) == "0.1,-3.5"
    assert nfx.centroid_key(16777216.0, 1e-7) == "16777216,0.0000001"
    assert nfx.centroid_key(float("nan"), 1.5) == "NaN,1.5"
    assert nfx.centroids_to_key_strings([[1.25, 2.0]]) == ["1.25,2"]


def test_no_cpu_fallback_without_gpu(nfx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nfx.NfxError):
        nfx.Context(0)


def test_column_dtypes_follow_the_reference_frame(nfx):
    """shape.rs:116-127: the seven polygon columns are geometric_features' f64 return values, every other column of
    every set was read out of an f32 tensor."""
    assert nfx.column_dtypes(0) == ["f64", "f32", "f32", "f32", "f32", "f64", "f64", "f64", "f32", "f64", "f64", "f64"]
    for s in range(1, 5):
        assert set(nfx.column_dtypes(s)) == {"f32"}
    assert nfx.lib().nfx_feature_set_column_dtype(0, 12) == -1 and nfx.lib().nfx_feature_set_column_dtype(5, 0) == -1


def test_a_host_without_rccl_gets_unsupported_not_a_crash():
    """include/nfx.h: nfx_comm_* return NFX_ERR_UNSUPPORTED when librccl cannot be loaded.  NFX_RCCL_PATH names the one file that
    is tried; a fresh process (the loader runs once per process) pointed at a file that does not exist must come back with
    status 2 -- the first build of this path built its message from two dlerror() calls, the second of which returns NULL."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"L = C.CDLL({os.path.join(ROOT, 'nuclei-feature-extraction_amd', 'libnfx.so')!r})\n"
        "buf = C.create_string_buffer(128)\n"
        "st = [L.nfx_comm_unique_id(buf) for _ in range(2)]\n"   # the second call takes the cached failure
        "print('status', st)\n"
    )
    env = dict(os.environ, NFX_RCCL_PATH="/nonexistent/librccl-not-here.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    assert "status [2, 2]" in r.stdout, r.stdout


def test_one_hip_runtime_per_process_and_the_bad_order_is_refused():
    """torch first: libnfx binds the runtime torch already mapped (ONE libamdhip64 in /proc/self/maps).  libnfx first: a later `import torch`
    maps a second one -- Context() refuses that order with the cure in the message instead of leaving torch with "No HIP GPUs"."""
    import subprocess
    import sys
    pre = "import sys, importlib; sys.path.insert(0, %r); " % ROOT
    good = pre + ("import torch; nfx = importlib.import_module('nuclei-feature-extraction_amd'); nfx.lib(); m = nfx.hip_runtimes_mapped(); "
                  "print(len(m), 'torch' in m[0])")
    r = subprocess.run([sys.executable, "-c", good], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["1", "True"], r.stdout + r.stderr[-500:]
    bad = pre + ("nfx = importlib.import_module('nuclei-feature-extraction_amd'); nfx.lib(); m0 = nfx.hip_runtimes_mapped(); import torch\n"
                 "m = nfx.hip_runtimes_mapped(); print(len(m0), 'rocm' in m0[0], len(m))\n"
                 "try:\n    nfx.Context(0)\n    print('no error')\nexcept nfx.NfxError as e:\n    print('refused' if 'two HIP runtimes' in str(e) and 'Import torch BEFORE' in str(e) else str(e))")
    r = subprocess.run([sys.executable, "-c", bad], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["1", "True", "2", "refused"], r.stdout + r.stderr[-500:]
    free = pre + "nfx = importlib.import_module('nuclei-feature-extraction_amd'); nfx.lib(); print(len(nfx.hip_runtimes_mapped()), 'torch' in sys.modules)"
    r = subprocess.run([sys.executable, "-c", free], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["1", "False"], r.stdout + r.stderr[-500:]


def test_a_plain_c_host_links_the_library_and_rebuilds_the_reference_frame_header(nfx, tmp_path):
    """The boundary is a C ABI: a host written in C -- what a Rust `extern "C"` block compiles down to -- includes nfx.h,
    links libnfx.so and rebuilds the reference's frame header (args.rs:18-73 expansion, column names in order, the centroid key) without a GPU;
    a context on a box without one is an error status, not a crash."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "nfx.h"
int main(void) {
    const char *names[] = {"geometry", "texture"};
    int sets[8];
    int n = nfx_parse_feature_sets(names, 2, sets, 8);
    printf("version %d stamp %s sets %d:", nfx_version(), nfx_build_stamp(), n);
    int cols = 0;
    for (int i = 0; i < n; ++i) { printf(" %s", nfx_feature_set_name(sets[i])); cols += nfx_feature_set_num_columns(sets[i]); }
    char buf[128];
    nfx_feature_set_column_name(sets[1], 0, buf, sizeof buf);
    printf("\ncolumns %d first-glcm %s", cols, buf);
    nfx_centroid_key(2047.96875f, 100.0f, buf, sizeof buf);
    printf(" key %s\n", buf);
    nfx_ctx *ctx = NULL;
    nfx_status st = nfx_ctx_create(0, &ctx);
    printf("ctx_create %d %s\n", (int)st, ctx ? "ctx" : "null");
    if (ctx) nfx_ctx_destroy(ctx);
    return 0;
}
''')
    exe = tmp_path / "host"
    lib_dir = os.path.dirname(nfx.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib_dir, "-lnfx",
                           f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    lines = out.stdout.splitlines()
    assert lines[0] == f"version 500 stamp {nfx.build_stamp()} sets 4: geometry GLCM GLRLM gabor filter"
    assert lines[1] == "columns 400 first-glcm correlation_0_1_32 key 2047.9688,100"
    assert lines[2].startswith("ctx_create ") and (lines[2].endswith(" ctx") == lines[2].startswith("ctx_create 0 "))
