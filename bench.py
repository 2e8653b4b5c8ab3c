#!/usr/bin/env python3
"""bench.py -- nuclei/sec of the per-nucleus feature path on MI355X (BASELINE.json metric: "nuclei/sec (full feature set)").

    python bench.py --gpus N --steps K --warmup W [--workload config5|config3|config2] [--inputs streamed|resident]

DEFAULT = the configuration the metric is quoted on, BASELINE.json configs[4]'s per-GPU shard: 1.25 M nuclei, 64 x 64 patches, the full
feature set (`all`, F = 418), batch 100, THE SLIDE STREAMED -- the nuclei's slide (47 703^2 pixels, 6.83 GB) and their rings start in
pinned host memory and every step streams them in band by band, overlapped with compute (nfx_extract_tile_streamed); a step is timed
from the first H2D byte to the last feature row in host memory (SURVEY.md §8d's wall-time definition).  `--inputs resident` (the
default of config3 / config2) times the same hot path with packed patches and rings resident in HBM.

A "step" is one pass of the hot path (polygon preprocessing + rasterisation + every selected feature set) over one rank's nuclei.
Synthetic inputs are generated on the GPU before the timed region (SURVEY.md §8d generator).

N > 1 (`--gpus N`): started bare, this script launches its own N ranks as child processes (torch.distributed.run, one per GPU) before
it touches a GPU itself; started by torch.distributed.run it is one of those ranks.  Nuclei are index-sharded on reference-batch
boundaries, no data-path collective during extraction.  Top line = WEAK scaling of the streamed shard: every rank
streams its own 1.25 M-nucleus slide over its own host link and its kernels write its rows into ONE host matrix shared by the
ranks (the reference's vstack under a mutex, main.rs:96-108, as N parallel PCIe streams: no gather step is left when the
destination is host memory).  The RCCL gather over xGMI (rows resident in HBM, gathered to rank 0: distributed.ShardedExtraction)
is measured in the same run: `also.weak_config3_rccl_gather`, `also.strong_scaling_config4` (BASELINE configs[3], gathered matrix
checked bit for bit against rank 0's own single-GPU rows) and `also.full_feature_set_resident`.

The one JSON line also carries
  roofline     -- the dominant kernel's algorithmic bytes / its average launch time (HIP events on the streams the kernels run on,
                  measured live in this process) against the HBM peak (`hbm`), the unit that really limits it from the committed
                  counter summary when that was taken on this csrc/ (`lds` incl. conflict_free_frac / `issue`), the dominant
                  kernel's HBM traffic and `traffic_step`: the whole step's traffic against its algorithmic bytes;
  cpu_baseline -- kind "aten-twin": the reference's libtorch-CPU tensor programs restated op for op on torch-CPU (tests/_torch_twin.py,
                  one process per core, batch 100; a bounded prefix, extrapolated); `cpu_baseline_port`: the C/OpenMP oracle port
                  (oracle/nfx_oracle.c) on a bounded sample -- both on this host's cores, rank 0 at N = 1 only;
  also         -- N = 1: configs[2] resident (the previous rounds' headline) with its own roofline, configs[2] with its slide
                  streamed, the full feature set resident with its own roofline.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "nuclei-feature-extraction_amd"

WORKLOADS = {
    # name: (nuclei per GPU, patch size, feature sets, BASELINE.json config it is, seed index, default inputs)
    "config2": (100_000, 32, ["geometry"], "configs[1]: 100k nuclei, 32x32, geometry", 1, "resident"),
    "config3": (1_000_000, 48, ["geometry", "color", "glcm"], "configs[2]: 1M nuclei, 48x48, geometry+color+glcm", 2, "resident"),
    "config5": (1_250_000, 64, ["all"], "configs[4] per-GPU shard: 1.25M nuclei, 64x64, all", 4, "streamed"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles, 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 0.5 * 2.4e9   # wave-instructions per second (nominal)
# what back-to-back independent plain FMAs really reach on this chip (scripts/valu_probe.hip, profiles/NOTEBOOK_r1-r5.md §6: 6.7e11 of the nominal
# 1.23e12, i.e. ~4 cycles per wave64 instruction and SIMD): the second basis `issue.frac_of_probe` is quoted against
VALU_ISSUE_PROBE = 6.7e11
LDS_CU_CYCLES_PEAK = 256 * 2.4e9          # CU-cycles per second: one LDS pipe per CU
# which unit limits a kernel when its counters are at hand; everything else on this path is VALU-issue-bound
LDS_BOUND_KERNELS = ("k_glcm",)
PMC_NUCLEI = {"config2": 100_000, "config3": 1_000_000, "config5": 1_250_000}   # what scripts/profile_round.sh profiles (one launch per step)


def csrc_stamp():
    """The hash of the kernel and ABI sources ON DISK (csrc/stamp.py -- the computation csrc/Makefile compiles into the library as
    nfx_build_stamp()): committed counter summaries carry the stamp they were taken on."""
    spec = importlib.util.spec_from_file_location("_nfx_stamp", os.path.join(ROOT, PKG, "csrc", "stamp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_stamp()


def library_stamp_on_disk(path=None):
    """The stamp compiled into a libnfx.so, read out of the FILE (the marker string "NFX_BUILD_STAMP=<stamp>", csrc/nfx_stamp.cpp): no child
    process, and a stale library is never mapped into this one (it could not be unloaded before the rebuilt one is loaded)."""
    path = path or os.path.join(ROOT, PKG, "libnfx.so")
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    at = blob.find(b"NFX_BUILD_STAMP=")
    if at < 0:
        return None
    end = blob.find(b"\0", at)
    return blob[at + 16:end].decode(errors="replace") if end > at else None


def library_is_fresh():
    """Whether the prebuilt libnfx.so was built from the sources on disk."""
    return library_stamp_on_disk() == csrc_stamp()


def require_fresh_library(nfx, source_stamp=None):
    """The binary that is about to be measured IS a build of the sources next to it, or no line is printed: nfx_build_stamp() (compiled
    in by csrc/Makefile) against the hash of the sources on disk.  Returns the stamp that goes into config.csrc_stamp."""
    built, src = nfx.build_stamp(), (source_stamp or csrc_stamp())
    if os.environ.get("NFX_LIB_PATH"):
        # an explicitly named alternative build (kernel A/Bs: scripts/lib_exp.py): measured as what it is -- its own stamp goes into the line,
        # next to the path, and the counters committed for the tree's sources are not applied to it
        return built + " (NFX_LIB_PATH=" + os.environ["NFX_LIB_PATH"] + ")"
    if built != src:
        sys.exit(f"bench.py: libnfx.so is stale -- built from sources with stamp {built}, the sources on disk hash to {src}; "
                 f"run `make -C {os.path.join(PKG, 'csrc')}` (no line is printed for a binary that is not a build of these sources)")
    return built


def committed_counters(workload):
    """(every timing slot's counter summary of the workload, source) from profiles/pmc_current.json when it was taken on this build of
    csrc/, else (None, why)."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None, "no profiles/pmc_current.json"
    if doc.get("csrc_stamp") != csrc_stamp():
        return None, f"profiles/pmc_current.json was taken on csrc {doc.get('csrc_stamp')}, this is {csrc_stamp()}"
    slots = (doc.get(workload) or {}).get("slots")
    if not slots:
        return None, f"profiles/pmc_current.json holds no counters for {workload}"
    return slots, doc.get("source", "profiles/pmc_current.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config5", choices=sorted(WORKLOADS))
    ap.add_argument("--inputs", default="auto", choices=["auto", "streamed", "resident"],
                    help="streamed (default of config5, the metric's configuration): slide and rings start in pinned host memory, every step streams "
                         "the slide in band by band and leaves the rows in host memory (first H2D byte -> last row on the host); resident "
                         "(default of config3 / config2): packed patches + rings in HBM before the timed region, rows left in HBM")
    ap.add_argument("--nuclei", type=int, default=0, help="nuclei per GPU (default: the workload's)")
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--sets", default="", help="override the workload's feature sets (comma separated; experiments only)")
    ap.add_argument("--gather-slices", type=int, default=4, help="N>1, resident legs: slices per step whose row gather overlaps the next slice's compute")
    ap.add_argument("--chunk", type=int, default=0, help="nuclei per streamed chunk (0: the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements of the `also` object")
    ap.add_argument("--strong-nuclei", type=int, default=0,
                    help="N>1: total nuclei of the strong-scaling (configs[3]) measurement; default: 1M when the top line runs at full size, else skipped")
    ap.add_argument("--also-nuclei", type=int, default=0,
                    help="nuclei per GPU of the secondary legs (config3 resident / streamed, full set resident); default: each workload's own "
                         "size when the top line runs at full size, else skipped")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of each CPU baseline's sample")
    ap.add_argument("--gather", default="rank0", choices=["rank0", "none"],
                    help="N>1, resident legs: rank0 (default) = every rank's rows are gathered to rank 0 inside the timed region (the reference's "
                         "vstack, main.rs:96-108); none = rows stay on the rank that computed them")
    ap.add_argument("--transport", default="torch", choices=["torch", "nfx"],
                    help="N>1, resident legs: what carries the rows to rank 0 -- torch.distributed.gather (c10d over RCCL; default) or the engine's "
                         "own nfx_gather_rows (include/nfx.h: grouped ncclSend / ncclRecv below any framework; has only ever run with one rank -- "
                         "tests/test_comm_gpu.py needs two GPUs for more -- so unverified for N>1)")
    ap.add_argument("--numa", default="auto", choices=["auto", "off"],
                    help="auto (default): before its first GPU call every rank binds itself to the CPUs of its GPU's NUMA node (sysfs), so that its "
                         "pinned slide, row matrix and staging are first-touched locally; recorded per rank in config.numa.  off: leave placement to the OS")
    ap.add_argument("--ipc-legacy", default="env", choices=["env", "0", "1"],
                    help="N>1: HSA_ENABLE_IPC_MODE_LEGACY for the ranks (RCCL peer access across processes).  env (default) = whatever the "
                         "environment holds, 0 if it holds nothing (dmabuf IPC: what this pool's host driver supports); recorded in the line")
    return ap.parse_args()


# ---- inputs -------------------------------------------------------------------------------------------------------------------------
class Resident:
    """One workload's synthetic inputs resident in HBM (nuclei [i0, i0+n) of the workload's seed stream)."""

    def __init__(self, nfx, synth, torch, dev, workload, n, i0, set_names):
        self.ps = WORKLOADS[workload][1]
        self.sets = nfx.parse_feature_sets(set_names)
        self.F = sum(nfx.num_columns(s) for s in self.sets)
        self.seed = synth.SEED_BASE + WORKLOADS[workload][4]
        self.n, self.i0 = n, i0
        t0 = time.time()
        self.off, self.xy = synth.polygons(self.seed, i0, n, self.ps)
        self.d_off = torch.from_numpy(self.off.astype(np.int32)).to(dev)
        self.d_off64 = self.d_off.to(torch.int64)
        self.d_xy = torch.from_numpy(self.xy).to(dev)
        self.pixel_sets = any(s != nfx.GEOMETRY for s in self.sets)
        self.d_patches = synth.patches(self.seed, i0, n, self.ps, xp=torch, device=dev) if self.pixel_sets else None
        torch.cuda.synchronize()
        self.gen_s = time.time() - t0
        self.torch = torch
        v_mean = float(len(self.xy)) / max(1, n)
        self.bytes_per_nucleus = (3 * self.ps * self.ps if self.pixel_sets else 0) + 8.0 * v_mean + 8 + 4 * self.F

    def extract(self, ctx, b0, b1, out, batch_size):
        """rows of the local nuclei [b0, b1) into out"""
        if b0 == 0 and b1 == self.n:
            ctx.extract(self.sets, self.d_off, self.d_xy, self.d_patches, self.ps, batch_size=batch_size, out=out, want_centroids=False)
            return
        v0, v1 = int(self.off[b0]), int(self.off[b1])
        off = (self.d_off64[b0:b1 + 1] - v0).to(self.torch.int32)
        ctx.extract(self.sets, off, self.d_xy[v0:v1], self.d_patches[b0:b1] if self.d_patches is not None else None, self.ps,
                    batch_size=batch_size, out=out, want_centroids=False)


class StreamedSlide:
    """BASELINE configs[4]'s input path for one rank: nuclei [i0, i0+n) and THEIR slide (side ceil(sqrt(n (ps/1.5)^2)); nuclei in
    slide order, as segmentation tools emit them) in pinned host memory, plus the rows of the same slide RESIDENT in HBM
    (nfx_extract_tile) that every streamed step has to reproduce bit for bit."""

    def __init__(self, nfx, synth, torch, ctx, dev, workload, n, i0, B, set_names=None):
        _, self.ps, names, self.desc, sidx, _ = WORKLOADS[workload]
        self.sets = nfx.parse_feature_sets(set_names or names)
        self.F = sum(nfx.num_columns(s) for s in self.sets)
        self.n, self.B = n, B
        ps = self.ps
        seed = synth.SEED_BASE + sidx
        self.T = T = synth.tile_side(n, ps)
        t0 = time.time()
        cx, cy = synth.centres_slide_order(seed, i0, n, n, ps, T, T, slide_i0=i0)
        self.off, self.xy = synth.polygons(seed, i0, n, ps, cx, cy)
        d_off = torch.from_numpy(self.off.astype(np.int32)).to(dev)
        d_xy = torch.from_numpy(self.xy).to(dev)
        cen, _ = ctx.preprocess_polygons(d_off, d_xy)          # the centroids the engine will address the windows with
        cen = cen.cpu().numpy()
        d_p = synth.patches(seed, i0, n, ps, xp=torch, device=dev)
        tile = synth.tile_from_patches(torch, d_p, cen[:, 0], cen[:, 1], T, T, ps, seed, dev)
        del d_p
        torch.cuda.empty_cache()
        ctx.reserve(self.sets, n, len(self.xy), ps, batch_size=B)
        ref, _ = ctx.extract_tile(self.sets, d_off, d_xy, tile, ps, batch_size=B)     # the slide resident in HBM: the rows to reproduce
        torch.cuda.synchronize()
        self.ref = ref.cpu().numpy()
        del ref
        self.h_tile = nfx.pinned_empty((3, T, T), np.uint8)
        for r0 in range(0, T, 2048):
            self.h_tile[:, r0:r0 + 2048] = tile[:, r0:r0 + 2048].cpu().numpy()
        del tile, d_off, d_xy
        torch.cuda.empty_cache()
        self.gen_s = time.time() - t0
        self.bytes_per_nucleus = 3 * ps * ps + 8.0 * len(self.xy) / max(1, n) + 8 + 4 * self.F   # SURVEY.md §8d (the packed-patch unit)
        self.stats = {}

    def step(self, ctx, rows, chunk=0):
        self.stats = ctx.extract_tile_streamed(self.sets, self.off, self.xy, self.h_tile, self.ps, batch_size=self.B, chunk_nuclei=chunk, out=rows)[2]

    def describe(self, el, steps, rows):
        st, n, ps, T = self.stats, self.n, self.ps, self.T
        wire = st["image_bytes_h2d"] + st["other_bytes_h2d"] + st["rows_bytes_d2h"]
        packed = 3 * ps * ps + 8.0 * len(self.xy) / n + 4 + 4 * self.F + 8          # what a packed-patch stream would move per nucleus
        return {"slide": f"{T} x {T} pixels, {3 * T * T / 1e9:.2f} GB, nuclei in slide order, pinned host memory; streamed every step in bands of 2048 rows "
                         "(contiguous copies overlapped with compute, nfx_extract_tile_streamed); timed from the first H2D byte to the last row on the host "
                         "(SURVEY.md §8d)",
                "rows_equal_resident_tile_rows": bool(np.array_equal(np.asarray(rows).view(np.uint32), self.ref.view(np.uint32))),
                "rows_written_by_the_kernels_into_host_memory": bool(st.get("rows_direct", 0)),
                "pcie_bytes_per_nucleus": round(wire / n, 1), "pcie_bytes_per_nucleus_packed_patch_stream": round(packed, 1),
                "link_GBps": round(wire * steps / el / 1e9, 2), "bands_copied": st["blocks_copied"], "bands_total": st["blocks_total"],
                "chunks": st["chunks"], "image_pinned": bool(st["image_pinned"])}


# ---- timing and the roofline object ----------------------------------------------------------------------------------------------------
def timed(torch, ctl, ctx, steps, warmup, step, fence):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
    Returns (seconds, per-kernel HIP-event timing of rank-local launches)."""
    for _ in range(warmup):
        step()
    fence()
    ctx.enable_timing(True)
    ctx.reset_timing()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.kernel_timing()
    ctx.enable_timing(False)
    if ctl is not None:
        el = torch.tensor([elapsed], dtype=torch.float64)
        ctl.all_reduce(el, op=ctl.ReduceOp.MAX)     # the control group is gloo: a host tensor
        elapsed = float(el.item())
    return elapsed, timing


def roofline_of(timing, steps, n, bytes_per_nucleus, workload, full_size):
    """The dominant kernel against the unit that limits it.  `hbm` (always present) is the figure the metric asks for: algorithmic bytes
    per launch / average launch time / 8 TB/s.  When the committed counter summary was taken on this very csrc/, the top-level bound /
    achieved / peak / frac describe the unit that really saturates first -- the LDS pipe for k_glcm, VALU issue for the rest -- else they
    fall back to the HBM figure and say so.  `traffic` is the dominant kernel's counted HBM traffic per launch, `traffic_step` every
    kernel's of one step against the step's algorithmic bytes (every pixel kernel re-reads the packed patch: DESIGN.md §6)."""
    live = {k: v for k, v in timing.items() if v["launches"]}
    dom_name, dom_t = max(live.items(), key=lambda kv: kv[1]["total_ms"])
    dom_avg_ms = dom_t["total_ms"] / dom_t["launches"]
    launches_per_step = max(1, dom_t["launches"] // max(1, steps))
    nuclei_per_launch = n / launches_per_step        # a kernel launched once per step (or once per streamed chunk) processes that many of the rank's nuclei
    sec = dom_avg_ms * 1e-3
    achieved = bytes_per_nucleus * nuclei_per_launch / sec / 1e9
    hbm = {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
           "algorithmic_bytes_per_nucleus": round(bytes_per_nucleus, 1), "nuclei_per_launch": round(nuclei_per_launch, 1),
           "launches_per_step": launches_per_step}
    slots, src = (None, "not the full-size workload") if not full_size else committed_counters(workload)
    issue = lds = traffic = traffic_step = None
    if slots is not None:
        per_nucleus = 1.0 / PMC_NUCLEI[workload]
        ctr = slots.get(dom_name)
        if ctr is not None:
            if ctr.get("traffic_bytes_per_launch") is not None:
                traffic = int(ctr["traffic_bytes_per_launch"] * per_nucleus * nuclei_per_launch)
            valu = ctr.get("valu_insts_per_nucleus")
            if valu is not None:
                rate = valu * nuclei_per_launch / sec
                issue = {"unit": "valu", "insts_per_nucleus": valu, "achieved": round(rate / 1e9, 2), "peak": round(VALU_ISSUE_PEAK / 1e9, 1),
                         "rate_unit": "G wave-instructions/s", "frac": round(rate / VALU_ISSUE_PEAK, 4),
                         "basis": "nominal: 256 CUs x 4 SIMDs x one wave64 instruction per 2 cycles x 2.4 GHz",
                         "frac_of_probe": round(rate / VALU_ISSUE_PROBE, 4),
                         "probe_basis": "6.7e11 wave-instructions/s: what scripts/valu_probe.hip reaches with back-to-back independent plain FMAs "
                                        "(~4 cycles per wave64 instruction and SIMD)"}
            lds_cyc = ctr.get("lds_idx_active_cycles_per_nucleus")
            if lds_cyc is not None:
                busy = lds_cyc * nuclei_per_launch / sec
                same = ctr.get("lds_addr_conflict_over_idx_active")
                lds = {"unit": "lds", "active_cycles_per_nucleus": lds_cyc, "achieved": round(busy / 1e9, 2), "peak": round(LDS_CU_CYCLES_PEAK / 1e9, 1),
                       "rate_unit": "G CU-cycles/s with the LDS pipe active (SQ_LDS_IDX_ACTIVE)", "frac": round(busy / LDS_CU_CYCLES_PEAK, 4),
                       "same_address_conflict_share": same, "bank_conflict_share": ctr.get("lds_bank_conflict_over_idx_active"),
                       # the pipe's busy cycles that move data: busy x (1 - the share spent serialising lanes of one instruction on one address)
                       "conflict_free_frac": None if same is None else round(busy / LDS_CU_CYCLES_PEAK * (1.0 - same), 4)}
        tot = sum(s.get("traffic_bytes_per_launch") or 0 for s in slots.values()) * per_nucleus * n
        alg = bytes_per_nucleus * n
        traffic_step = {"bytes": int(tot), "algorithmic_bytes": int(alg), "ratio": round(tot / alg, 3),
                        "what": "HBM bytes of EVERY kernel of one step (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, the resident-input profile of this "
                                "workload scaled to this step's nuclei) against the step's algorithmic bytes: every pixel-consuming kernel reads the packed "
                                "patch again (k_hue_batch, k_color, k_glcm, k_glrlm, k_gabor)"}
    top = dict(hbm, bound="hbm")
    if dom_name in LDS_BOUND_KERNELS and lds is not None:
        top = {"bound": "lds", "achieved": lds["achieved"], "peak": lds["peak"], "unit": lds["rate_unit"], "frac": lds["frac"]}
    elif issue is not None:
        top = {"bound": "valu", "achieved": issue["achieved"], "peak": issue["peak"], "unit": issue["rate_unit"], "frac": issue["frac"]}
    note = ("VALU issue + LDS atomics (DESIGN.md §3): no kernel of this path is HBM-bound at these byte counts; `hbm` is reported because "
            "the metric asks for it" + ("; a streamed step launches every kernel once per chunk and runs two chunks at a time on two streams: a launch's "
                                        "duration here includes what the other stream's kernels took from it (see `exclusive`)" if launches_per_step > 1 else ""))
    if top["bound"] == "hbm":
        note += "; bound says hbm only because no counter summary of this csrc/ is committed (counters_source)"
    return {"bound": top["bound"], "kernel": dom_name, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
            "frac": top["frac"], "traffic": traffic, "traffic_step": traffic_step, "hbm": hbm, "issue": issue, "lds": lds,
            "counters_source": src, "limiter": note, "avg_launch_ms": round(dom_avg_ms, 4),
            "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 4) for k, v in live.items()}}


# ---- CPU baselines --------------------------------------------------------------------------------------------------------------------
def aten_twin_baseline(args, workload, set_names):
    """BASELINE.md §3's "restatement of the libtorch-CPU path": tests/_twin_baseline.py as a child process, run to completion BEFORE this
    process touches a GPU (its workers are forked CPU processes)."""
    import subprocess
    _, ps, _, _, sidx, _ = WORKLOADS[workload]
    cores = int(os.environ.get("NFX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_twin_baseline.py"), "--ps", str(ps), "--nuclei", str(args.nuclei or WORKLOADS[workload][0]),
           "--batch", str(args.batch_size), "--procs", str(cores), "--seconds", str(min(args.cpu_seconds, 15.0)), "--seed-index", str(sidx),
           "--sets", ",".join(set_names)]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        d = json.loads(line[-1]) if out.returncode == 0 and line else {"error": (out.stderr or "no output")[-300:]}
    except Exception as e:   # a reported baseline, never a reason for the bench to fail
        d = {"error": repr(e)[:300]}
    if "value" in d:
        d["sample"] += " -- a prefix of the workload, the rate extrapolated to the whole of it"
    return d


def oracle_port_baseline(args, nfx, sets, off, xy, patches_u8_of, ps, n_local, got_rows):
    """The C/OpenMP oracle port on a bounded sample of the same workload; doubles as an end-of-bench parity spot check.
    patches_u8_of(m) -> numpy u8 [m,3,ps,ps] of the first m nuclei; got_rows(m) -> the GPU's rows of the first m nuclei (numpy)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc      # checker / baseline only -- never on the product path
    import _parity as par
    B = args.batch_size
    # on the CPUs the process started with (the NUMA prologue confined it to its GPU's node; the baseline is a host measurement): the
    # OpenMP workers are created inside the first call and inherit the mask in force then
    bound_to = os.sched_getaffinity(0)
    os.sched_setaffinity(0, getattr(args, "cpu_affinity0", None) or bound_to)
    cores = int(os.environ.get("NFX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))   # the GPU box grants 16 host cores per GPU

    def run_sample(m):
        m = min(max(B, (m // B) * B), n_local)
        pu = patches_u8_of(m)
        t = time.perf_counter()
        rows, _ = orc.extract(sets, off[:m + 1], xy[:off[m]], pu, ps, batch_size=B, threads=cores)
        return m, time.perf_counter() - t, rows

    try:
        m0, dt0, _ = run_sample(max(B, cores * B))
        m1, dt1, rows = run_sample(int(m0 / dt0 * args.cpu_seconds))
    finally:
        os.sched_setaffinity(0, bound_to)
    worst = par.assert_rows_close(got_rows(m1), rows, par.column_scales(nfx, sets, ps), what="bench sample vs oracle")
    return {"value": round(m1 / dt1, 1), "unit": "nuclei/s", "cores": cores, "kind": "port",
            "what": "C/OpenMP oracle port of the reference's arithmetic (oracle/nfx_oracle.c): a stronger (faster) CPU baseline than the reference's "
                    "libtorch-CPU path, whose tensor programs `cpu_baseline` (kind aten-twin) restates",
            "sample": f"first {m1} nuclei of the same workload, OpenMP over reference batches ({cores} threads), {dt1:.1f} s",
            "parity_spot_check_worst_scaled_err": float(f"{worst:.3g}")}


# ---- ranks ----------------------------------------------------------------------------------------------------------------------------
class Ranks:
    """The process group(s) of an N > 1 run.  The CONTROL group is always gloo (barriers, timing reductions, checksums, agreement on
    failures: host memory, cannot be taken down by a GPU transport problem); the DATA group carries the row gathers: RCCL (backend
    "nccl") when every rank has a GPU of its own and RCCL came up on ALL ranks, else the gloo group through host memory (a rehearsal
    on fewer GPUs than ranks, or a degraded run that says so in the line)."""

    def __init__(self, torch, dist, dev, world, rank, rehearsal):
        import datetime
        self.dist, self.world, self.rank = dist, world, rank
        self.rccl_failure = None
        self.rccl_ranks_seen = None     # what the probe all-reduce of ones returned on this rank: the number of ranks RCCL really joined
        self.wire_cpu = rehearsal
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        self.data = None            # None = the default (gloo) group
        if not rehearsal:
            ok, why = 1, ""
            try:
                grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180), device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe, group=grp)               # the first collective brings the communicator up: a failure shows here
                torch.cuda.synchronize()
                self.rccl_ranks_seen = int(probe.item())
                if self.rccl_ranks_seen != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
            except Exception as e:
                ok, why = 0, repr(e)[:300]
                print(f"bench.py: RCCL failed on rank {rank} ({why})", file=sys.stderr, flush=True)
            # every rank learns whether ALL ranks have a working communicator, over the control group: nobody switches on its own
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.data = grp
            else:
                whys = [None] * world
                dist.all_gather_object(whys, why)
                self.rccl_failure = next((w for w in whys if w), "a rank failed")
                self.wire_cpu = True
                if ok:
                    try:
                        dist.destroy_process_group(grp)
                    except Exception as e:   # a half-built communicator may refuse: the run goes on over gloo
                        print(f"bench.py: destroying the RCCL group failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
                if rank == 0:
                    print(f"bench.py: RCCL did not come up on every rank ({self.rccl_failure}); the row gathers fall back to gloo through host memory "
                          "(the line says degraded: true)", file=sys.stderr, flush=True)
        dist.barrier()

    def barrier(self):
        self.dist.barrier()

    def close(self):
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass


class SharedRows:
    """ONE [world * n, F] f32 matrix in host memory that every rank maps (a file in /dev/shm named after the rendezvous port) and whose
    own slice every rank registers with its HIP runtime: the kernels of rank r write rank r's rows straight into rows [r n, (r+1) n)
    over rank r's own host link.  The reference's row merge (vstack_mut under a mutex, main.rs:96-108) without a gather step."""

    def __init__(self, nfx, ranks, n, F):
        self.nfx, self.ranks = nfx, ranks
        self.path = f"/dev/shm/nfx_rows_{os.environ.get('MASTER_PORT', '0')}_{ranks.world}"
        shape = (ranks.world * n, F)
        err = ""
        if ranks.rank == 0:
            try:
                need = 4 * shape[0] * shape[1] + 4096
                vfs = os.statvfs("/dev/shm")       # a tmpfs file may be created larger than the space behind it: writing it then raises SIGBUS, so ask first
                if vfs.f_bavail * vfs.f_frsize < need * 1.05:
                    raise OSError(f"/dev/shm has {vfs.f_bavail * vfs.f_frsize >> 20} MiB free, the shared row matrix needs {need >> 20} MiB")
                np.lib.format.open_memmap(self.path, mode="w+", dtype=np.float32, shape=shape).flush()
            except OSError as e:
                err = repr(e)
        errs = [None] * ranks.world
        ranks.dist.all_gather_object(errs, err)
        self.full = self.mine = None
        self.registered = False
        if any(errs):
            self.why_not = next(e for e in errs if e)
            return
        self.full = np.load(self.path, mmap_mode="r+")
        self.mine = self.full[ranks.rank * n:(ranks.rank + 1) * n]
        self.registered = nfx.lib().nfx_host_register(self.mine.ctypes.data, self.mine.nbytes) == 0
        self.why_not = None if self.registered else "hipHostRegister refused the shared mapping (rows staged on the device and copied home instead)"

    def close(self):
        if self.mine is not None and self.registered:
            self.nfx.lib().nfx_host_unregister(self.mine.ctypes.data)
        self.mine = self.full = None
        self.ranks.barrier()
        if self.ranks.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.ipc_legacy != "env":
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = args.ipc_legacy     # explicit: recorded in the line as asked
    elif args.gpus > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC, the only mode this pool's host driver supports
    if args.gpus > 1 and world_env is None:
        # bare `python bench.py --gpus N`: start the N ranks as children BEFORE this process touches a GPU
        D = importlib.import_module(PKG + ".distributed")
        sys.exit(D.launch_ranks(os.path.abspath(__file__), args.gpus, sys.argv[1:]))
    n_def, ps, set_names, wl_desc, _, default_inputs = WORKLOADS[args.workload]
    if args.sets:
        set_names = args.sets.split(",")
        wl_desc += f" [feature sets overridden: {args.sets}]"
    inputs = default_inputs if args.inputs == "auto" else args.inputs
    # the libtorch-shaped CPU baseline: a child process, finished before any GPU call here
    cpu_twin = None
    if int(world_env or "1") == 1 and not args.no_cpu_baseline:
        cpu_twin = aten_twin_baseline(args, args.workload, set_names)

    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not library_is_fresh():
        # a checkout without the (git-ignored) built library, or one whose library was built from other sources: build it as the driver's
        # build step would (make rebuilds what changed).  Local rank 0 builds, the others wait; the product itself never builds or falls
        # back at run time, and require_fresh_library below refuses to measure a binary that still is not a build of these sources
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not library_is_fresh() and time.time() - t_wait < 900:
                time.sleep(2.0)
    nfx = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    D = importlib.import_module(PKG + ".distributed")
    built_stamp = require_fresh_library(nfx)

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()          # (counts devices without initialising the runtime)
    # NUMA: bind this rank to its GPU's node BEFORE the first GPU call and the first pinned allocation (distributed.py).  The CPU
    # baselines keep the process's original CPU set (cpu_affinity0)
    cpu_affinity0 = sorted(os.sched_getaffinity(0))
    numa = {"bound": False, "why": "--numa off"} if args.numa == "off" else D.bind_rank_to_gpu_numa(local_rank % max(1, n_dev))
    args.cpu_affinity0 = cpu_affinity0
    # fewer GPUs than ranks (a one-GPU box rehearsing the N>1 path): ranks share devices and the collectives run over gloo
    # through host memory -- same sharding, slicing and verification code, NOT a scaling measurement (flagged in the line)
    rehearsal = world > max(1, n_dev)
    torch.cuda.set_device(local_rank % max(1, n_dev))
    dev = torch.device("cuda", local_rank % max(1, n_dev))
    torch.cuda.init()
    # the collective path is taken for N > 1 (or when forced, to rehearse RCCL itself with one rank)
    use_dist = world > 1 or (os.environ.get("NFX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    ranks = None
    if use_dist:
        # RCCL announces itself on stdout ("Librccl path : ...") when its first communicator comes up; stdout carries the
        # one JSON line of the contract, so the process's fd 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            ranks = Ranks(torch, dist, dev, world, rank, rehearsal)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    ctl = dist if use_dist else None
    wire_cpu = ranks.wire_cpu if ranks else False
    rccl_failure = ranks.rccl_failure if ranks else None
    data_group = ranks.data if ranks else None

    B = args.batch_size
    n_local = max(B, ((args.nuclei or n_def) // B) * B)          # shard boundary on a reference-batch boundary
    full_size = n_local == n_def and not args.sets
    sets = nfx.parse_feature_sets(set_names)
    F = sum(nfx.num_columns(s) for s in sets)
    pixel_sets = any(s != nfx.GEOMETRY for s in sets)
    if inputs == "streamed" and not pixel_sets:
        inputs = "resident"          # geometry needs no pixels: nothing to stream
    ctx = nfx.Context(dev.index)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    def fence(sx=None):
        if sx is not None:
            sx.finish()
        torch.cuda.synchronize()      # this rank's device work is done ...
        if use_dist:
            dist.barrier()            # ... and so is every other rank's (the control group's barrier is a host barrier)
        torch.cuda.synchronize()

    def sharded(n_total, Fx):
        return D.ShardedExtraction(n_total, Fx, B, n_slices=args.gather_slices, device=dev, group=data_group,
                                   wire_device="cpu" if wire_cpu else None, transport="torch" if wire_cpu else args.transport)

    also = {}
    cfg_extra = {}
    cfg_rccl = {}
    cpu_port = None
    # what every rank's NUMA prologue found and did, in rank order (config.numa)
    numa_all = [numa]
    if use_dist:
        numa_all = [None] * world
        dist.all_gather_object(numa_all, dict(numa, rank=rank, cpus_allowed_at_start=len(cpu_affinity0)))
    # ---- the top line ----------------------------------------------------------------------------------------------------------------
    if inputs == "streamed":
        slide = StreamedSlide(nfx, synth, torch, ctx, dev, args.workload, n_local, rank * n_local, B, set_names)
        gen_s, bytes_per_nucleus = slide.gen_s, slide.bytes_per_nucleus
        shared = None
        if use_dist:
            shared = SharedRows(nfx, ranks, n_local, F)
        rows = shared.mine if (shared is not None and shared.mine is not None) else nfx.pinned_empty((n_local, F), np.float32)
        elapsed, timing = timed(torch, ctl, ctx, args.steps, args.warmup, lambda: slide.step(ctx, rows, args.chunk), fence)
        cfg_extra = slide.describe(elapsed, args.steps, rows)
        if use_dist:
            # every rank's streamed rows, as they lie in the ONE host matrix, against the rows of that rank's slide resident in HBM
            mine = torch.tensor([D.row_checksum(torch.from_numpy(slide.ref))], dtype=torch.int64)
            got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, got, dst=0)
            fence()
            if rank == 0:
                src = shared.full if shared.full is not None else None
                cfg_extra["rows_destination"] = (f"one [{world * n_local}, {F}] f32 matrix in host memory shared by the ranks ({shared.path}): every rank's kernels write "
                                                 "its shard over its own host link" if src is not None else
                                                 f"per-rank pinned host buffers (no shared matrix: {shared.why_not})")
                if src is not None:
                    cfg_extra["shared_matrix_registered_for_direct_writes"] = shared.registered
                    cfg_extra["every_ranks_rows_in_the_shared_matrix_equal_its_resident_rows"] = bool(all(
                        D.row_checksum(torch.from_numpy(np.ascontiguousarray(src[r * n_local:(r + 1) * n_local]))) == int(got[r].item()) for r in range(world)))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            h = slide
            # the oracle port on the patches the engine itself gathers for the first nuclei (nfx_gather_patches: utils.rs:159-191)
            cen = ctx.preprocess_polygons(h.off, h.xy)[0]

            def patches_of(m):
                out = np.empty((m, 3, ps, ps), np.uint8)
                for i0 in range(0, m, 2000):      # windows of the first m nuclei lie in the slide's first bands
                    i1 = min(m, i0 + 2000)
                    rmax = int(np.ceil(cen[i0:i1, 1].max())) + ps
                    out[i0:i1] = ctx.gather_patches(cen[i0:i1], np.ascontiguousarray(h.h_tile[:, :rmax]), ps)
                return out
            cpu_port = oracle_port_baseline(args, nfx, sets, h.off, h.xy, patches_of, ps, n_local, lambda m: np.asarray(rows[:m]))
        if shared is not None:
            rows = None
            shared.close()
        del slide
        torch.cuda.empty_cache()
        res = None
    else:
        res = Resident(nfx, synth, torch, dev, args.workload, n_local, rank * n_local, set_names)
        gen_s, bytes_per_nucleus = res.gen_s, res.bytes_per_nucleus
        ctx.reserve(sets, n_local, len(res.xy), ps, batch_size=B)   # device-memory calls never allocate after this
        d_out = torch.empty((n_local, F), dtype=torch.float32, device=dev)
        sx = None
        if use_dist and args.gather == "rank0":
            sx = sharded(world * n_local, F)
            assert (sx.i0, sx.i1) == (rank * n_local, (rank + 1) * n_local)

            def step():
                sx.step(lambda b0, b1, out: res.extract(ctx, b0, b1, out, B))
        else:
            def step():
                res.extract(ctx, 0, n_local, d_out, B)
        elapsed, timing = timed(torch, ctl, ctx, args.steps, args.warmup, step, lambda: fence(sx))
        if sx is not None:
            ok = sx.verify_transport()
            if rank == 0:
                also["gather_checksums_match"] = bool(all(ok))
            d_out = sx.local_rows()
            if not args.no_also:
                # the same shards with the rows left where they were computed: what the extraction alone scales to
                el_c, _ = timed(torch, ctl, ctx, args.steps, 1, lambda: res.extract(ctx, 0, n_local, d_out, B), fence)
                if rank == 0:
                    also["weak_scaling_compute_only"] = {
                        "value": round(world * n_local * args.steps / el_c, 1), "unit": "nuclei/s", "ms_per_step": round(el_c / args.steps * 1e3, 3),
                        "scaling": "weak", "what": "the top line's shards extracted with NO gather (rows stay on the rank that computed them)",
                        "gather_cost_ms_per_step": round((elapsed - el_c) / args.steps * 1e3, 3)}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu_port = oracle_port_baseline(args, nfx, sets, res.off, res.xy,
                                            lambda m: res.d_patches[:m].cpu().numpy() if res.pixel_sets else np.zeros((m, 3, ps, ps), np.uint8),
                                            ps, n_local, lambda m: d_out[:m].cpu().numpy())
        cfg_extra = {"inputs": "packed u8 patches + polygons resident in HBM, rows left in HBM"}
    value = world * n_local * args.steps / elapsed
    roofline = roofline_of(timing, args.steps, n_local, bytes_per_nucleus, args.workload, full_size)

    # ---- also: the other figures, under the same clock ------------------------------------------------------------------------------------
    def resident_leg(workload, n, k, with_gather):
        """one workload with resident inputs on every rank (rows gathered to rank 0 over the data group for N > 1): (entry, Resident)"""
        names = WORKLOADS[workload][2]
        r = Resident(nfx, synth, torch, dev, workload, n, rank * n, names)
        ctx.reserve(r.sets, r.n, len(r.xy), r.ps, batch_size=B)
        if use_dist and with_gather:
            sxl = sharded(world * n, r.F)
            el, t = timed(torch, ctl, ctx, k, 1, lambda: sxl.step(lambda b0, b1, out: r.extract(ctx, b0, b1, out, B)), lambda: fence(sxl))
            okl = sxl.verify_transport()
            # the same shards with the rows left on the rank that computed them: what the gather adds to a step
            o = sxl.local_rows()
            el_nogather, _ = timed(torch, ctl, ctx, k, 1, lambda: r.extract(ctx, 0, r.n, o[:r.n], B), fence)
        else:
            o = torch.empty((r.n, r.F), dtype=torch.float32, device=dev)
            el, t = timed(torch, ctl, ctx, k, 1, lambda: r.extract(ctx, 0, r.n, o, B), fence)
            okl = None
        e = {"value": round(world * r.n * k / el, 1), "unit": "nuclei/s", "n_gpus": world, "steps": k, "ms_per_step": round(el / k * 1e3, 3), "scaling": "weak",
             "config": {"workload": WORKLOADS[workload][3], "nuclei_per_gpu": r.n, "patch_size": r.ps, "feature_sets": names, "columns": r.F, "batch_size": B,
                        "inputs": "packed u8 patches + polygons resident in HBM" + (", rows gathered to rank 0 " + ("(gloo through host memory)" if wire_cpu else "(RCCL)")
                                                                                    if okl is not None else ", rows left in HBM"),
                        "gen_seconds": round(r.gen_s, 1)},
             "roofline": roofline_of(t, k, r.n, r.bytes_per_nucleus, workload, n == WORKLOADS[workload][0])}
        if okl is not None:
            e["gather_checksums_match"] = bool(all(okl)) if rank == 0 else None
            e["gather"] = {"transport": "gloo through host memory" if wire_cpu else ("RCCL (torch.distributed.gather)" if args.transport == "torch" else "RCCL (nfx_gather_rows)"),
                           "bytes_per_step_to_rank0": 4 * (world - 1) * r.n * r.F, "ms_per_step_without_gather": round(el_nogather / k * 1e3, 3),
                           "cost_ms_per_step": round((el - el_nogather) / k * 1e3, 3)}
        return e, r

    na = lambda wl: ((args.also_nuclei or (WORKLOADS[wl][0] if full_size else 0)) // B) * B
    if not args.no_also and not args.sets and args.workload == "config5" and inputs == "streamed":
        k2 = max(2, args.steps // 2)
        if na("config3") > 0:
            e, r3 = resident_leg("config3", na("config3"), args.steps, True)
            if rank == 0:
                also["config3_resident" if world == 1 else "weak_config3_rccl_gather"] = e
                if world > 1 and "gather" in e:
                    # on the TOP LEVEL of an N > 1 line (the driver's parsed record keeps `config`): the row gather over RCCL / xGMI that
                    # north_star names, measured in this same run on configs[2]'s shards (the top line's own rows land in host memory)
                    cfg_rccl = {"rccl_gather_ms_per_step": e["ms_per_step"], "rccl_gather_cost_ms_per_step": e["gather"]["cost_ms_per_step"],
                                "rccl_gather_bytes_per_step": e["gather"]["bytes_per_step_to_rank0"], "rccl_gather_transport": e["gather"]["transport"],
                                "rccl_gather_checksums_match": e["gather_checksums_match"],
                                "rccl_gather_workload": f"{WORKLOADS['config3'][3]}, {e['config']['nuclei_per_gpu']} nuclei per GPU, rows gathered to rank 0"}
            # BASELINE configs[3]: the SAME nuclei cut into `world` batch-aligned shards (strong scaling); rank 0's weak-scaling inputs start with
            # exactly those nuclei, so it can check the gathered matrix against its own single-GPU result
            n_strong = args.strong_nuclei or (WORKLOADS["config3"][0] if full_size else 0)     # (any number: the last shard may be ragged, later ones empty)
            if world > 1 and use_dist and n_strong > 0:
                ssx = sharded(n_strong, r3.F)
                rs = Resident(nfx, synth, torch, dev, "config3", ssx.n_local, ssx.i0, WORKLOADS["config3"][2])
                ctx.reserve(rs.sets, max(rs.n, n_strong if rank == 0 else 0), max(len(rs.xy), len(r3.xy)) + 64 * n_strong, rs.ps, batch_size=B)
                el_s, _ = timed(torch, ctl, ctx, args.steps, 1, lambda: ssx.step(lambda b0, b1, out: rs.extract(ctx, b0, b1, out, B)), lambda: fence(ssx))
                oks = ssx.verify_transport()
                full = ssx.result()
                if rank == 0:
                    whole = r3 if n_strong <= r3.n else Resident(nfx, synth, torch, dev, "config3", n_strong, 0, WORKLOADS["config3"][2])
                    single = torch.empty((n_strong, r3.F), dtype=torch.float32, device=dev)
                    whole.extract(ctx, 0, n_strong, single, B)
                    torch.cuda.synchronize()
                    also["strong_scaling_config4"] = {
                        "value": round(n_strong * args.steps / el_s, 1), "unit": "nuclei/s", "ms_per_step": round(el_s / args.steps * 1e3, 3), "scaling": "strong",
                        "what": f"BASELINE.json configs[3]: the first {n_strong} nuclei of {WORKLOADS['config3'][3]} index-sharded x{world} (batch-aligned shards), "
                                "rows gathered to rank 0", "shard_nuclei": ssx.counts, "gather_checksums_match": bool(all(oks)),
                        "gathered_equals_single_gpu": bool(torch.equal(full.to(dev).view(torch.int32), single.view(torch.int32))),
                        "gathered_checksum": D.row_checksum(full)}
                    single = whole = None
                dist.barrier()
                ssx = rs = full = None
            r3 = None
            torch.cuda.empty_cache()
            if world == 1:
                s3 = StreamedSlide(nfx, synth, torch, ctx, dev, "config3", na("config3"), 0, B)
                hr = nfx.pinned_empty((s3.n, s3.F), np.float32)
                el3, _ = timed(torch, None, ctx, k2, 1, lambda: s3.step(ctx, hr, args.chunk), fence)
                also["config3_streamed"] = dict({"value": round(s3.n * k2 / el3, 1), "unit": "nuclei/s", "ms_per_step": round(el3 / k2 * 1e3, 3),
                                                 "workload": WORKLOADS["config3"][3]}, **s3.describe(el3, k2, hr))
                del s3, hr
                torch.cuda.empty_cache()
        if na("config5") > 0:
            e, r5 = resident_leg("config5", na("config5"), k2, True)
            r5 = None
            torch.cuda.empty_cache()
            if rank == 0:
                also["full_feature_set_resident"] = e
                # The streamed step runs its chunks two at a time on two streams, so a launch's HIP-event duration there includes what the other
                # stream's kernels took from it (k_gabor: ~11.5 ms per 89 k-nucleus chunk launch against 5.2 ms alone).  The SAME kernel's launch
                # with the GPU to itself -- this leg: same nuclei, same feature sets, resident inputs, one launch per step -- is put beside it
                rx = e["roofline"]
                roofline["exclusive"] = {"what": "the same workload with resident inputs (also.full_feature_set_resident): one launch per kernel and step, nothing "
                                                 "running beside it -- the figure that speaks for the kernel; the top-level one is the streamed step's own (launches of "
                                                 "two chunks overlap on two streams)",
                                         "kernel": rx["kernel"], "bound": rx["bound"], "achieved": rx["achieved"], "peak": rx["peak"], "unit": rx["unit"], "frac": rx["frac"],
                                         "avg_launch_ms": rx["avg_launch_ms"], "hbm_frac": rx["hbm"]["frac"]}

    if rank == 0:
        sharding = "single GPU"
        if world > 1:
            how = "REHEARSAL on fewer GPUs than ranks, not a scaling measurement" if rehearsal else ("RCCL FAILED to come up: " + str(rccl_failure) if rccl_failure else "")
            if inputs == "streamed":
                sharding = f"index-sharded x{world} on reference-batch boundaries; every rank streams its own slide and leaves its rows in host memory" + (f" ({how})" if how else "")
            elif args.gather == "none":
                sharding = f"index-sharded x{world}, NO gather: rows stay on the rank that computed them" + (f" ({how})" if how else "")
            else:
                sharding = f"index-sharded x{world}, rows gathered to rank 0 (" + ("gloo through host memory: " + how if wire_cpu else "RCCL") + ")"
        line = {
            "metric": "nuclei/sec", "value": round(value, 1), "unit": "nuclei/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64 (u8 pixels, integer counts)", "data": "synthetic",
            "config": dict({"workload": wl_desc + (", slide streamed from pinned host memory, rows to host memory" if inputs == "streamed" else ""),
                            "nuclei_per_gpu": n_local, "patch_size": ps, "feature_sets": set_names, "columns": F, "batch_size": B,
                            "inputs_mode": inputs, "sharding": sharding, "gen_seconds": round(gen_s, 1), "csrc_stamp": built_stamp,
                            "hip_runtimes_mapped": nfx.hip_runtimes_mapped(),
                            "gather": (None if world == 1 else ("none needed: rows land in host memory" if inputs == "streamed" else args.gather)),
                            "transport": (("torch.distributed.gather" if (wire_cpu or args.transport == "torch") else "nfx_gather_rows (grouped ncclSend/ncclRecv)")
                                          if use_dist and inputs != "streamed" and args.gather == "rank0" else None),
                            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") if use_dist else None,
                            "rccl_failure": rccl_failure,
                            "rccl_ranks_seen": ranks.rccl_ranks_seen if ranks else None,   # the probe all-reduce of ones: how many ranks RCCL joined
                            "numa": numa_all}, **cfg_rccl, **cfg_extra),
            "roofline": roofline,
            # TWO CPU baselines, each under its own name; `cpu_baseline` (the contract's key) repeats the one that restates the reference's
            # own libtorch-CPU tensor programs (kind aten-twin) when it was measured, else the port -- and says which in `is`
            "cpu_baseline": dict(cpu_twin, **{"is": "cpu_baseline_aten_twin"}) if cpu_twin is not None else (dict(cpu_port, **{"is": "cpu_baseline_port"}) if cpu_port else None),
            "cpu_baseline_aten_twin": cpu_twin, "cpu_baseline_port": cpu_port,
            "also": also or None,
        }
        gpu1 = value / world     # per-GPU rate against the host's cores
        line["speedup_vs_cpu"] = {k: round(gpu1 / b["value"], 1) for k, b in (("cpu_baseline_aten_twin", cpu_twin), ("cpu_baseline_port", cpu_port))
                                  if b and b.get("value")} or None
        if rccl_failure:
            line["degraded"] = True     # a multi-GPU figure whose rows travelled through host memory over gloo: not an xGMI measurement
        print(json.dumps(line), flush=True)
    if ranks:
        ranks.close()
    ctx.close()


if __name__ == "__main__":
    main()
