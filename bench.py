#!/usr/bin/env python3
"""bench.py -- nuclei/sec of the per-nucleus feature path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload config3|config2|config5]

A "step" is one pass of the hot path (polygon preprocessing + rasterisation + every selected feature
set, rows gathered on rank 0) over one resident batch of synthetic nuclei.  Inputs are generated
directly in HBM before the timed region (SURVEY.md §8d generator).

Default workload = BASELINE.json configs[2], the largest single-GPU configuration:
1M nuclei, 48x48 patches, geometry+color+glcm (F=254), batch 100.

N > 1 (`--gpus N`): started bare, this script launches its own N ranks as child processes
(torch.distributed.run, one per GPU) before it touches a GPU itself; started by torch.distributed.run it is
one of those ranks.  The headline is WEAK scaling -- every rank holds its own index shard of `--nuclei`
nuclei (shard boundaries are multiples of the reference batch size), extracts it with no
data-path collective and its rows are gathered to rank 0 over RCCL (the reference's vstack under a mutex,
main.rs:96-108) through `distributed.ShardedExtraction`, the class the CPU gloo test drives.  The same run
then times BASELINE.json configs[3] (the 1M nuclei of configs[2] cut into N shards: strong scaling) and rank 0
checks that the gathered matrix is bit-identical to its own single-GPU extraction of the same nuclei; both
go into the line's `also` object.

The one JSON line also carries
  roofline     -- the dominant kernel's algorithmic bytes / its average launch time (HIP events on the
                  engine's stream, measured live in this process) against the HBM peak, plus the VALU issue
                  rate of the same kernel from the committed SQ counter summary (if taken on this csrc/);
  cpu_baseline -- the CPU oracle (oracle/nfx_oracle.c, a C/OpenMP port of the reference's arithmetic)
                  timed on this host on a bounded sample of the same workload;
  also         -- N = 1: the full-feature-set rate of the configs[4] per-GPU shard (the metric's "full feature
                  set") with its own roofline, and the PCIe-inclusive rate of the headline workload (patches
                  streamed from pinned host memory, rows copied back to pinned host memory, all inside the
                  timed region: SURVEY.md §8d's wall-time definition).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "nuclei-feature-extraction_amd"

WORKLOADS = {
    # name: (nuclei per GPU, patch size, feature sets, BASELINE.json config it is, seed index)
    "config2": (100_000, 32, ["geometry"], "configs[1]: 100k nuclei, 32x32, geometry", 1),
    "config3": (1_000_000, 48, ["geometry", "color", "glcm"], "configs[2]: 1M nuclei, 48x48, geometry+color+glcm", 2),
    "config5": (1_250_000, 64, ["all"], "configs[4] per-GPU shard: 1.25M nuclei, 64x64, all", 4),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles, 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 0.5 * 2.4e9   # wave-instructions per second (nominal)
# what back-to-back independent plain FMAs really reach on this chip (scripts/valu_probe.hip, DESIGN.md §6: 6.7e11 of the nominal
# 1.23e12, i.e. ~4 cycles per wave64 instruction and SIMD): the second basis `issue.frac_of_probe` is quoted against
VALU_ISSUE_PROBE = 6.7e11
LDS_CU_CYCLES_PEAK = 256 * 2.4e9          # CU-cycles per second: one LDS pipe per CU
# which unit limits a kernel when its counters are at hand; everything else on this path is VALU-issue-bound
LDS_BOUND_KERNELS = ("k_glcm",)


def csrc_stamp():
    """sha256 over the kernel and ABI sources: committed counter summaries carry the stamp they were taken on."""
    d = os.path.join(ROOT, PKG, "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(workload, kernel):
    """(counter summary of the workload's dominant kernel, source) from profiles/pmc_current.json when it was taken on this build
    of csrc/ for this workload and kernel, else (None, why)."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None, "no profiles/pmc_current.json"
    if doc.get("csrc_stamp") != csrc_stamp():
        return None, f"profiles/pmc_current.json was taken on csrc {doc.get('csrc_stamp')}, this is {csrc_stamp()}"
    t = (doc.get(workload) or {}).get("slots", {}).get(kernel)
    if not t:
        return None, f"profiles/pmc_current.json holds no counters for {workload}/{kernel}"
    return t, doc.get("source", "profiles/pmc_current.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--nuclei", type=int, default=0, help="nuclei per GPU (default: the workload's)")
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--sets", default="", help="override the workload's feature sets (comma separated; experiments only)")
    ap.add_argument("--h2d", default="resident", choices=["resident", "stream"],
                    help="resident (default, the metric): inputs in HBM before the timed region; stream: patches start in pinned "
                         "host memory, every step streams them in and the rows back out (PCIe-inclusive rate, DESIGN.md)")
    ap.add_argument("--gather-slices", type=int, default=4, help="N>1: slices per step whose row gather overlaps the next slice's compute")
    ap.add_argument("--chunk", type=int, default=100_000, help="nuclei per streamed chunk (--h2d stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements of the `also` object")
    ap.add_argument("--strong-nuclei", type=int, default=0,
                    help="N>1: total nuclei of the strong-scaling (configs[3]) measurement; default: the workload's size when the headline "
                         "runs at full size, else skipped")
    ap.add_argument("--full-set-nuclei", type=int, default=0,
                    help="nuclei per GPU of the full-feature-set (configs[4] shard) measurement; default: 1.25M when the headline runs at "
                         "full size, else skipped")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--gather", default="rank0", choices=["rank0", "none"],
                    help="N>1: rank0 (default) = every rank's rows are gathered to rank 0 inside the timed region (the reference's vstack, "
                         "main.rs:96-108); none = rows stay on the rank that computed them (compute-only scaling).  With rank0 the line also "
                         "carries `also.weak_scaling_compute_only`, measured in the same run")
    ap.add_argument("--transport", default="torch", choices=["torch", "nfx"],
                    help="N>1: what carries the rows to rank 0 -- torch.distributed.gather (c10d over RCCL; default) or the engine's own "
                         "nfx_gather_rows (include/nfx.h: grouped ncclSend / ncclRecv below any framework; torch.distributed only hands out the "
                         "communicator id; has only ever run with one rank -- tests/test_comm_gpu.py needs two GPUs for more -- so unverified for N>1)")
    ap.add_argument("--ipc-legacy", default="env", choices=["env", "0", "1"],
                    help="N>1: HSA_ENABLE_IPC_MODE_LEGACY for the ranks (RCCL peer access across processes).  env (default) = whatever the "
                         "environment holds, 0 if it holds nothing (dmabuf IPC: what this pool's host driver supports); recorded in the line")
    return ap.parse_args()


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


def timed(torch, dist, use_dist, ctx, steps, warmup, step, fence):
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
    if use_dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        if dist.get_backend() == "gloo":
            el = el.cpu()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    return elapsed, timing


def roofline_of(timing, steps, res, workload, full_size):
    """The dominant kernel against the unit that limits it.  `hbm` (always present) is the figure the metric asks for: algorithmic
    bytes per launch / average launch time / 8 TB/s.  When the committed counter summary was taken on this very csrc/, the top-level
    bound / achieved / peak / frac describe the unit that really saturates first -- the LDS pipe for k_glcm, VALU issue for the
    rest -- else they fall back to the HBM figure and say so."""
    dom_name, dom_t = max(timing.items(), key=lambda kv: kv[1]["total_ms"])
    dom_avg_ms = dom_t["total_ms"] / max(1, dom_t["launches"])
    launches_per_step = max(1, dom_t["launches"] // max(1, steps))
    nuclei_per_launch = res.n / launches_per_step    # a kernel launched once per step processes the rank's nuclei
    sec = dom_avg_ms * 1e-3
    achieved = res.bytes_per_nucleus * nuclei_per_launch / sec / 1e9
    hbm = {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
           "algorithmic_bytes_per_nucleus": round(res.bytes_per_nucleus, 1)}
    ctr, src = (None, "not the full-size workload") if not full_size else committed_counters(workload, dom_name)
    issue = lds = None
    traffic = None
    if ctr is not None:
        traffic = ctr.get("traffic_bytes_per_launch")
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
            lds = {"unit": "lds", "active_cycles_per_nucleus": lds_cyc, "achieved": round(busy / 1e9, 2), "peak": round(LDS_CU_CYCLES_PEAK / 1e9, 1),
                   "rate_unit": "G CU-cycles/s with the LDS pipe active (SQ_LDS_IDX_ACTIVE)", "frac": round(busy / LDS_CU_CYCLES_PEAK, 4),
                   "same_address_conflict_share": ctr.get("lds_addr_conflict_over_idx_active"),
                   "bank_conflict_share": ctr.get("lds_bank_conflict_over_idx_active")}
    top = dict(hbm, bound="hbm")
    if dom_name in LDS_BOUND_KERNELS and lds is not None:
        top = {"bound": "lds", "achieved": lds["achieved"], "peak": lds["peak"], "unit": lds["rate_unit"], "frac": lds["frac"]}
    elif issue is not None:
        top = {"bound": "valu", "achieved": issue["achieved"], "peak": issue["peak"], "unit": issue["rate_unit"], "frac": issue["frac"]}
    note = ("VALU issue + LDS atomics (DESIGN.md §3): no kernel of this path is HBM-bound at these byte counts; `hbm` is reported because "
            "the metric asks for it")
    if top["bound"] == "hbm":
        note += "; bound says hbm only because no counter summary of this csrc/ is committed (counters_source)"
    return {"bound": top["bound"], "kernel": dom_name, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
            "frac": top["frac"], "traffic": traffic, "hbm": hbm, "issue": issue, "lds": lds, "counters_source": src, "limiter": note,
            "avg_launch_ms": round(dom_avg_ms, 4),
            "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 4) for k, v in timing.items() if v["launches"]}}


def streamed_tile_leg(nfx, synth, torch, dist, ctx, dev, workload, n, i0, B, steps, fence):
    """BASELINE configs[4]'s input path on one GPU: this rank's nuclei [i0, i0+n) and THEIR slide (side ceil(sqrt(n (ps/1.5)^2)),
    SURVEY.md §8d; nuclei in slide order, as segmentation tools emit them) start in pinned host memory; every step streams the
    slide block by block through nfx_extract_tile_streamed -- copies overlapped with compute, rows copied back -- and is timed from the
    first H2D byte to the last row on the host.  Checked bit for bit against nfx_extract_tile on the same slide resident in HBM."""
    n_def, ps, set_names, desc, sidx = WORKLOADS[workload]
    sets = nfx.parse_feature_sets(set_names)
    F = sum(nfx.num_columns(s) for s in sets)
    seed = synth.SEED_BASE + sidx
    T = synth.tile_side(n, ps)
    t0 = time.time()
    cx, cy = synth.centres_slide_order(seed, i0, n, n, ps, T, T, slide_i0=i0)
    off, xy = synth.polygons(seed, i0, n, ps, cx, cy)
    d_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    d_xy = torch.from_numpy(xy).to(dev)
    cen, _ = ctx.preprocess_polygons(d_off, d_xy)          # the centroids the engine will address the windows with
    cen = cen.cpu().numpy()
    d_p = synth.patches(seed, i0, n, ps, xp=torch, device=dev)
    tile = synth.tile_from_patches(torch, d_p, cen[:, 0], cen[:, 1], T, T, ps, seed, dev)
    del d_p
    torch.cuda.empty_cache()
    ctx.reserve(sets, n, len(xy), ps, batch_size=B)
    ref, _ = ctx.extract_tile(sets, d_off, d_xy, tile, ps, batch_size=B)     # the slide resident in HBM: the rows to reproduce
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    h_tile = nfx.pinned_empty((3, T, T), np.uint8)         # pinned for libnfx's own HIP runtime (torch's pinned memory is not)
    for r0 in range(0, T, 2048):
        h_tile[:, r0:r0 + 2048] = tile[:, r0:r0 + 2048].cpu().numpy()
    del tile
    torch.cuda.empty_cache()
    h_rows = nfx.pinned_empty((n, F), np.float32)
    gen_s = time.time() - t0
    stats = {}

    def step():
        stats.update(ctx.extract_tile_streamed(sets, off, xy, h_tile, ps, batch_size=B, out=h_rows)[2])
    el, _ = timed(torch, dist, False, ctx, steps, 1, step, fence)
    same = bool(np.array_equal(h_rows.view(np.uint32), ref.view(np.uint32)))
    wire = stats["image_bytes_h2d"] + stats["other_bytes_h2d"] + stats["rows_bytes_d2h"]
    packed = 3 * ps * ps + 8.0 * len(xy) / n + 4 + 4 * F + 8          # what the packed-patch stream moves per nucleus
    return {"value": round(n * steps / el, 1), "unit": "nuclei/s", "ms_per_step": round(el / steps * 1e3, 3),
            "what": f"{desc}: the nuclei's slide ({T} x {T} pixels, {3 * T * T / 1e9:.2f} GB, nuclei in slide order) starts in pinned host memory and is "
                    "streamed every step in bands of 2048 rows (contiguous copies overlapped with compute, nfx_extract_tile_streamed), rows copied back; timed from the "
                    "first H2D byte to the last row on the host (SURVEY.md §8d); never the headline value",
            "rows_equal_resident_tile_rows": same, "pcie_bytes_per_nucleus": round(wire / n, 1),
            "pcie_bytes_per_nucleus_packed_patch_stream": round(packed, 1), "link_GBps": round(wire * steps / el / 1e9, 2),
            "blocks_copied": stats["blocks_copied"], "blocks_total": stats["blocks_total"], "chunks": stats["chunks"], "gen_seconds": round(gen_s, 1)}


def aten_twin_baseline(args, ps, n):
    """BASELINE.md §3's "restatement of the libtorch-CPU path" for the geometry workload: tests/_twin_baseline.py as a child
    process, run to completion BEFORE this process touches a GPU (its workers are forked CPU processes)."""
    import subprocess
    cores = int(os.environ.get("NFX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_twin_baseline.py"), "--ps", str(ps), "--nuclei", str(n), "--batch", str(args.batch_size),
           "--procs", str(cores), "--seconds", str(min(args.cpu_seconds, 12.0))]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if out.returncode == 0 and line else {"error": (out.stderr or "no output")[-300:]}
    except Exception as e:   # a reported baseline, never a reason for the bench to fail
        return {"error": repr(e)[:300]}


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
    # the libtorch-shaped CPU data point of the geometry workload: a child process, finished before any GPU call here
    cpu_torch = None
    if args.workload == "config2" and int(world_env or "1") == 1 and not args.no_cpu_baseline and not args.sets:
        cpu_torch = aten_twin_baseline(args, WORKLOADS["config2"][1], args.nuclei or WORKLOADS["config2"][0])

    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, PKG, "libnfx.so")):
        # a checkout without the (git-ignored) built library: build it as the driver's build step would.  Rank 0 builds,
        # the others wait for the file; the product itself never builds or falls back at run time
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            while not os.path.exists(os.path.join(ROOT, PKG, "libnfx.so")):
                time.sleep(1.0)
            time.sleep(2.0)
    nfx = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    D = importlib.import_module(PKG + ".distributed")

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    # fewer GPUs than ranks (a one-GPU box rehearsing the N>1 path): ranks share devices and the collective runs over gloo
    # through host memory -- same sharding, slicing and verification code, NOT a scaling measurement (flagged in the line)
    rehearsal = world > max(1, n_dev)
    torch.cuda.set_device(local_rank % max(1, n_dev))
    dev = torch.device("cuda", local_rank % max(1, n_dev))
    # the collective path is taken for N > 1 (or when forced, to rehearse RCCL itself with one rank)
    use_dist = world > 1 or (os.environ.get("NFX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    wire_cpu = rehearsal          # the gathers travel through host memory (gloo): a rehearsal, or RCCL failed to come up
    rccl_failure = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL announces itself on stdout ("Librccl path : ...") when its first communicator comes up; stdout carries the
        # one JSON line of the contract, so the process's fd 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                try:
                    dist.init_process_group("nccl", device_id=dev)
                    probe = torch.ones(1, device=dev)
                    dist.all_reduce(probe)                      # the first collective brings the communicator up: a failure shows here
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
                except Exception as e:                          # RCCL did not come up (every rank sees the same environment): the scaling run
                    rccl_failure = repr(e)[:300]                # still measures sharding and compute, with the rows gathered over gloo through host
                    print(f"bench.py: RCCL failed on rank {rank} ({rccl_failure}); falling back to gloo through host memory", file=sys.stderr, flush=True)
                    if dist.is_initialized():
                        dist.destroy_process_group()
                    dist.init_process_group("gloo")
                    wire_cpu = True
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    n_def, ps, set_names, wl_desc, _ = WORKLOADS[args.workload]
    if args.sets:
        set_names = args.sets.split(",")
        wl_desc += f" [feature sets overridden: {args.sets}]"
    B = args.batch_size
    n_local = max(B, ((args.nuclei or n_def) // B) * B)          # shard boundary on a reference-batch boundary
    full_size = n_local == n_def and not args.sets

    res = Resident(nfx, synth, torch, dev, args.workload, n_local, rank * n_local, set_names)
    sets, F, gen_s = res.sets, res.F, res.gen_s
    ctx = nfx.Context(dev.index)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.reserve(sets, n_local, len(res.xy), ps, batch_size=B)   # device-memory calls never allocate after this
    d_out = torch.empty((n_local, F), dtype=torch.float32, device=dev)

    def fence(sx=None):
        if sx is not None:
            sx.finish()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the headline: resident inputs; N > 1 weak scaling through ShardedExtraction ----------------------------------
    stream_mode = args.h2d == "stream" and res.pixel_sets
    sx = None
    h_patches = h_rows = None
    if stream_mode:
        pipeline = importlib.import_module(PKG + ".pipeline")
        h_patches = torch.empty(res.d_patches.shape, dtype=torch.uint8, pin_memory=True)
        h_patches.copy_(res.d_patches)
        h_rows = torch.empty((n_local, F), dtype=torch.float32, pin_memory=True)
        torch.cuda.synchronize()

        def step():
            pipeline.extract_streamed(ctx, sets, res.off, res.xy, h_patches, ps, batch_size=B, chunk_nuclei=args.chunk, out=d_out,
                                      host_rows=h_rows, want_centroids=False)
    elif use_dist and args.gather == "rank0":
        transport = "torch" if wire_cpu else args.transport       # (the native gather needs one GPU per rank and a working RCCL)
        sx = D.ShardedExtraction(world * n_local, F, B, n_slices=args.gather_slices, device=dev, wire_device="cpu" if wire_cpu else None,
                                 transport=transport)
        assert (sx.i0, sx.i1) == (rank * n_local, (rank + 1) * n_local)

        def step():
            sx.step(lambda b0, b1, out: res.extract(ctx, b0, b1, out, B))
    else:
        def step():
            res.extract(ctx, 0, n_local, d_out, B)

    elapsed, timing = timed(torch, dist, use_dist, ctx, args.steps, args.warmup, step, lambda: fence(sx))
    total_nuclei = n_local * world
    value = total_nuclei * args.steps / elapsed
    roofline = roofline_of(timing, args.steps, res, args.workload, full_size)
    also = {}
    if sx is not None:
        ok = sx.verify_transport()
        if rank == 0:
            also["gather_checksums_match"] = bool(all(ok))
        d_out = sx.local_rows()

    if sx is not None and not args.no_also:
        # the same shards with the rows left where they were computed: if the incast into rank 0 (world-1 streams of n_local x F
        # x 4 B per step over its xGMI links) does not hide behind compute, this figure says what the extraction alone scales to
        el_c, _ = timed(torch, dist, True, ctx, args.steps, 1, lambda: res.extract(ctx, 0, n_local, d_out, B), fence)
        if rank == 0:
            also["weak_scaling_compute_only"] = {
                "value": round(total_nuclei * args.steps / el_c, 1), "unit": "nuclei/s", "ms_per_step": round(el_c / args.steps * 1e3, 3),
                "scaling": "weak", "what": "the headline's shards extracted with NO gather (rows stay on the rank that computed them): "
                "compute-only scaling, beside the gathered headline measured in the same run",
                "gather_cost_ms_per_step": round((elapsed - el_c) / args.steps * 1e3, 3)}

    # ---- CPU baseline: the oracle port on a bounded sample, rank 0 at N=1 only ------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as orc      # checker / baseline only -- never on the product path
        import _parity as par
        # the GPU box grants 16 host cores per GPU (use NFX_CPU_THREADS to override)
        cores = int(os.environ.get("NFX_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))

        def run_sample(m):
            m = min(max(B, (m // B) * B), n_local)
            pu = res.d_patches[:m].cpu().numpy() if res.pixel_sets else np.zeros((m, 3, ps, ps), np.uint8)
            so, sxy = res.off[:m + 1], res.xy[:res.off[m]]
            t = time.perf_counter()
            rows, _ = orc.extract(sets, so, sxy, pu, ps, batch_size=B, threads=cores)
            return m, time.perf_counter() - t, rows

        m0, dt0, _ = run_sample(max(B, cores * B))
        m1, dt1, rows = run_sample(int(m0 / dt0 * args.cpu_seconds))
        # the sample doubles as an end-of-bench parity spot check of the resident result
        got = (h_rows[:m1].numpy() if stream_mode else d_out[:m1].cpu().numpy())
        worst = par.assert_rows_close(got, rows, par.column_scales(nfx, sets, ps), what="bench sample vs oracle")
        cpu = {"value": round(m1 / dt1, 1), "unit": "nuclei/s", "cores": cores, "kind": "port",
               "what": "C/OpenMP oracle port of the reference's arithmetic (oracle/nfx_oracle.c), not the torch-CPU twin BASELINE.md §3 "
                       "describes: a stronger (faster) CPU baseline than the reference's libtorch-CPU path would be",
               "sample": f"first {m1} nuclei of the same workload, OpenMP over reference batches ({cores} threads), {dt1:.1f} s",
               "parity_spot_check_worst_scaled_err": float(f"{worst:.3g}")}

    # ---- also: the figures the metric names, under the same clock --------------------------------------------------------
    n_strong = args.strong_nuclei or (n_def if full_size else 0)
    n_strong = (n_strong // B) * B
    if not args.no_also and not stream_mode and world > 1 and use_dist and 0 < n_strong <= n_local:
        # configs[3]: the SAME n_strong nuclei cut into `world` batch-aligned shards (strong scaling); rank 0's weak-scaling
        # inputs start with exactly those nuclei, so it can check the gathered matrix against its own single-GPU result
        ssx = D.ShardedExtraction(n_strong, F, B, n_slices=args.gather_slices, device=dev, wire_device="cpu" if wire_cpu else None,
                                  transport="torch" if wire_cpu else args.transport)
        rs = Resident(nfx, synth, torch, dev, args.workload, ssx.n_local, ssx.i0, set_names)
        el_s, _ = timed(torch, dist, True, ctx, args.steps, 1, lambda: ssx.step(lambda b0, b1, out: rs.extract(ctx, b0, b1, out, B)),
                        lambda: fence(ssx))
        ok = ssx.verify_transport()
        full = ssx.result()
        if rank == 0:
            single = torch.empty((n_strong, F), dtype=torch.float32, device=dev)
            res.extract(ctx, 0, n_strong, single, B)
            torch.cuda.synchronize()
            same = bool(torch.equal(full.to(dev).view(torch.int32), single.view(torch.int32)))
            also["strong_scaling_config4"] = {
                "value": round(n_strong * args.steps / el_s, 1), "unit": "nuclei/s", "ms_per_step": round(el_s / args.steps * 1e3, 3),
                "scaling": "strong", "what": f"BASELINE.json configs[3]: the first {n_strong} nuclei of {wl_desc} index-sharded x{world} "
                "(batch-aligned shards), rows gathered to rank 0",
                "shard_nuclei": ssx.counts, "gather_checksums_match": bool(all(ok)), "gathered_equals_single_gpu": same,
                "gathered_checksum": D.row_checksum(full)}
            single = None
        dist.barrier()
        ssx = rs = full = None
    n5 = ((args.full_set_nuclei or (WORKLOADS["config5"][0] if full_size else 0)) // B) * B
    if not args.no_also and args.workload == "config3" and not args.sets and not stream_mode and (full_size or n5 > 0):
        if world == 1 and full_size:
            # (ii) PCIe-inclusive rate of the headline workload: first H2D byte to last row in pinned host memory
            pipeline = importlib.import_module(PKG + ".pipeline")
            hp = torch.empty(res.d_patches.shape, dtype=torch.uint8, pin_memory=True)
            hp.copy_(res.d_patches)
            hr = torch.empty((n_local, F), dtype=torch.float32, pin_memory=True)
            torch.cuda.synchronize()
            el_s, _ = timed(torch, dist, False, ctx, args.steps, 1,
                            lambda: pipeline.extract_streamed(ctx, sets, res.off, res.xy, hp, ps, batch_size=B, chunk_nuclei=args.chunk,
                                                              out=d_out, host_rows=hr, want_centroids=False), fence)
            same = bool(torch.equal(hr.view(torch.int32), d_out.cpu().view(torch.int32)))
            also["streamed_pcie_inclusive"] = {
                "value": round(n_local * args.steps / el_s, 1), "unit": "nuclei/s", "ms_per_step": round(el_s / args.steps * 1e3, 3),
                "what": f"{wl_desc}; patches start in pinned host memory and stream in every step in chunks of {args.chunk} nuclei "
                        "(double-buffered), rows are copied back to pinned host memory chunk by chunk; timed from the first H2D "
                        "byte to the last row on the host (SURVEY.md §8d); never the headline value",
                "host_rows_equal_resident_rows": same}
            del hp, hr
            # (iii) the same workload fed the way configs[4] asks: the TILE streamed block by block, not packed patches
            also["streamed_tile"] = streamed_tile_leg(nfx, synth, torch, dist, ctx, dev, "config3", n_local, 0, B, max(2, args.steps // 2), fence)
        # (i) the metric's "full feature set": configs[4] per-GPU shard on every rank (weak scaling, rows gathered to rank 0 for N > 1)
        res = d_out = sx = None
        torch.cuda.empty_cache()
        r5 = Resident(nfx, synth, torch, dev, "config5", n5, rank * n5, WORKLOADS["config5"][2])
        ctx.reserve(r5.sets, r5.n, len(r5.xy), r5.ps, batch_size=B)
        k5 = max(2, args.steps // 2)
        if use_dist:
            sx5 = D.ShardedExtraction(world * n5, r5.F, B, n_slices=args.gather_slices, device=dev, wire_device="cpu" if wire_cpu else None,
                                      transport="torch" if wire_cpu else args.transport)
            el5, t5 = timed(torch, dist, True, ctx, k5, 1, lambda: sx5.step(lambda b0, b1, out: r5.extract(ctx, b0, b1, out, B)), lambda: fence(sx5))
            ok5 = sx5.verify_transport()
        else:
            o5 = torch.empty((r5.n, r5.F), dtype=torch.float32, device=dev)
            el5, t5 = timed(torch, dist, False, ctx, k5, 1, lambda: r5.extract(ctx, 0, r5.n, o5, B), fence)
            ok5 = None
        if rank == 0:
            also["full_feature_set"] = {
                "value": round(world * r5.n * k5 / el5, 1), "unit": "nuclei/s", "n_gpus": world, "steps": k5, "ms_per_step": round(el5 / k5 * 1e3, 3),
                "scaling": "weak",
                "config": {"workload": WORKLOADS["config5"][3], "nuclei_per_gpu": r5.n, "patch_size": r5.ps, "feature_sets": ["all"],
                           "columns": r5.F, "batch_size": B, "gen_seconds": round(r5.gen_s, 1)},
                "roofline": roofline_of(t5, k5, r5, "config5", n5 == WORKLOADS["config5"][0])}
            if ok5 is not None:
                also["full_feature_set"]["gather_checksums_match"] = bool(all(ok5))
        r5 = o5 = None
        torch.cuda.empty_cache()
        if world == 1 and n5 == WORKLOADS["config5"][0]:
            # BASELINE configs[4] as it says: one GPU's shard of the 10 M-nucleus slide, full feature set, tile streamed over the host link
            also["streamed_tile_full_set"] = streamed_tile_leg(nfx, synth, torch, dist, ctx, dev, "config5", n5, 0, B, 2, fence)

    if rank == 0:
        sharding = "single GPU"
        if world > 1 and (stream_mode or args.gather == "none"):
            sharding = f"index-sharded x{world}, NO gather: rows stay on the rank that computed them" + (
                " (in its pinned host memory)" if stream_mode else "") + (" -- REHEARSAL on fewer GPUs than ranks" if rehearsal else "")
        elif world > 1:
            sharding = f"index-sharded x{world}, rows gathered to rank 0 " + ("(gloo through host memory: REHEARSAL on fewer GPUs than "
                                                                             "ranks, not a scaling measurement)" if rehearsal else
                                                                             (f"(gloo through host memory: RCCL FAILED to come up -- {rccl_failure})" if rccl_failure else "(RCCL)"))
        line = {
            "metric": "nuclei/sec", "value": round(value, 1), "unit": "nuclei/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64 (u8 pixels, integer counts)", "data": "synthetic",
            "config": {"workload": wl_desc, "nuclei_per_gpu": n_local, "patch_size": ps, "feature_sets": set_names,
                       "columns": F, "batch_size": B, "sharding": sharding,
                       "inputs": "packed u8 patches + polygons resident in HBM" if not stream_mode else
                                 f"patches streamed from pinned host memory every step in chunks of {args.chunk}, rows copied back to "
                                 "pinned host memory (PCIe-inclusive; NOT the headline metric)",
                       "gen_seconds": round(gen_s, 1), "csrc_stamp": csrc_stamp(),
                       "gather": ("none" if (stream_mode or args.gather == "none") else "rank0") if world > 1 else None,
                       "transport": (("torch.distributed.gather" if (wire_cpu or args.transport == "torch") else "nfx_gather_rows (grouped ncclSend/ncclRecv)")
                                     if use_dist and not stream_mode and args.gather == "rank0" else None),
                       "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") if use_dist else None,
                       "rccl_failure": rccl_failure},
            "roofline": roofline, "cpu_baseline": cpu, "also": also or None,
        }
        if cpu_torch is not None:
            line["cpu_baseline_torch"] = cpu_torch
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
