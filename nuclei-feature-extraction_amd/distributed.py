"""Index sharding of nuclei over ranks and the gather of feature rows.

The reference's multi-GPU mode is `gpus[rayon_thread_idx % gpu_count]` per batch plus a host-side
`vstack_mut` under a mutex (utils.rs:215-221, main.rs:96-108).  Here: one process per GPU, each rank
extracts a contiguous index shard whose boundaries are multiples of the reference batch size (the
hue statistics couple the nuclei of a batch, color.rs:144-155, so a batch must live on one rank),
with no data-path collective; the rows are then gathered to one rank (`torch.distributed`, backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

`ShardedExtraction` is the one implementation of that step: `bench.py` drives it with the HIP engine
as the per-slice extractor, `tests/test_distributed.py` drives it (gloo, world size 2) with the CPU
oracle in the engine's place.  `launch_ranks` starts one child process per GPU for a bare
`python bench.py --gpus N`.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, List, Optional, Sequence, Tuple


def shard_range(n: int, batch_size: int, rank: int, world: int) -> Tuple[int, int]:
    """Nuclei [i0, i1) of `rank`: rank k gets reference batches [k*ceil(nb/R), (k+1)*ceil(nb/R))."""
    if batch_size < 1 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    nb = (n + batch_size - 1) // batch_size
    per = (nb + world - 1) // world
    b0 = min(rank * per, nb)
    b1 = min((rank + 1) * per, nb)
    return min(b0 * batch_size, n), min(b1 * batch_size, n)


def shard_counts(n: int, batch_size: int, world: int) -> List[int]:
    return [shard_range(n, batch_size, r, world)[1] - shard_range(n, batch_size, r, world)[0] for r in range(world)]


def slice_bounds(rows: int, batch_size: int, n_slices: int) -> List[int]:
    """Cuts [0, rows) into at most n_slices pieces whose inner boundaries are multiples of batch_size."""
    nb = (rows + batch_size - 1) // batch_size
    n_slices = max(1, min(n_slices, nb))
    return [min(rows, (nb * k // n_slices) * batch_size) for k in range(n_slices)] + [rows]


def row_checksum(rows):
    """Order-sensitive checksum of a [m, F] f32 block from its bit patterns (NaN payloads included):
    sum over elements of bits * (1 + flat index mod 65521), in wrapping int64.  Tensor in, python int out."""
    import torch

    flat = rows.contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    w = torch.arange(flat.numel(), dtype=torch.int64, device=flat.device) % 65521 + 1
    return int((flat * w).sum().item())


class ShardedExtraction:
    """One rank's part of an index-sharded extraction of `n_total` nuclei and the gather of its rows to `dst`.

    Every rank extracts its shard [i0, i1) in a few batch-aligned slices; each slice's rows start their way to
    `dst` (one asynchronous gather per slice) while the next slice is computed, and two row buffers alternate
    between steps so the last gathers of step k overlap the first slices of step k+1.  Ragged shards are padded
    to the largest shard for the collective (the padding rows are never read).  Rows land on `dst` in rank, i.e.
    index, order: `result()` is the [n_total, F] matrix a single process would have produced.

    extract_fn(b0, b1, out): computes the LOCAL rows [b0, b1) (nuclei i0+b0 .. i0+b1) into `out` ([b1-b0, F]).
    wire_device: where the collective's buffers live (None = the compute device; "cpu" for a gloo group fed
                 from a GPU, the one-GPU rehearsal of bench.py).
    transport:   "torch" (default): torch.distributed.gather per slice (c10d over RCCL or gloo).
                 "nfx": the engine's own nfx_gather_rows (include/nfx.h: grouped ncclSend / ncclRecv below any framework, ragged
                 shards unpadded) on a side stream of its own, ordered after each slice by an event; torch.distributed only
                 carries the 128-byte communicator id to the ranks.
    """

    def __init__(self, n_total: int, F: int, batch_size: int, n_slices: int = 4, device=None, group=None, dst: int = 0,
                 wire_device=None, transport: str = "torch"):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.dst = dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total, self.F, self.batch_size = n_total, F, batch_size
        self.i0, self.i1 = shard_range(n_total, batch_size, self.rank, self.world)
        self.counts = shard_counts(n_total, batch_size, self.world)
        self.n_local = self.i1 - self.i0
        self.m = max(self.counts)                      # rows per rank in the collective
        self.bounds = slice_bounds(self.m, batch_size, n_slices)
        self.device = device if device is not None else torch.device("cpu")
        self.wire = torch.device(wire_device) if wire_device is not None else self.device
        self.bufs = [torch.empty((self.m, F), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.full = None
        self.blocks = None
        if self.rank == dst and transport == "torch":
            self.full = torch.empty((self.world * self.m, F), dtype=torch.float32, device=self.wire)
            self.blocks = [self.full[r * self.m:(r + 1) * self.m] for r in range(self.world)]
        self.pending: list = []
        self.steps = 0
        self.last_buf = None
        self.transport = transport
        if transport == "nfx":
            import importlib
            nfx = importlib.import_module(__package__)
            uid = [nfx.comm_unique_id() if self.rank == dst else None]
            dist.broadcast_object_list(uid, src=dst, group=group)
            self.cstream = torch.cuda.Stream(self.device)
            self.cctx = nfx.Context(self.device.index, stream=self.cstream.cuda_stream)   # a context of its own: its stream is the gathers' stream
            self.comm = nfx.Comm(self.cctx, self.rank, self.world, uid[0])
            # per slice: how many rows every rank really has in it (ragged shards travel unpadded), and the root's landing block
            nb = len(self.bounds) - 1
            self.slice_counts = [[max(0, min(self.bounds[k + 1], c) - self.bounds[k]) for c in self.counts] for k in range(nb)]
            self.slice_dst = [torch.empty((sum(sc), F), dtype=torch.float32, device=self.device) for sc in self.slice_counts] if self.rank == dst else [None] * nb
            self.buf_free = [None, None]   # event: the gathers that read bufs[b] have finished
        elif transport != "torch":
            raise ValueError("transport is 'torch' or 'nfx'")

    def _step_nfx(self, extract_fn):
        import torch
        b = self.steps & 1
        buf = self.bufs[b]
        compute = torch.cuda.current_stream(self.device)
        if self.buf_free[b] is not None:
            compute.wait_event(self.buf_free[b])        # the gathers of two steps ago still read this buffer
        for k in range(len(self.bounds) - 1):
            b0, b1 = self.bounds[k], self.bounds[k + 1]
            e1 = min(b1, self.n_local)
            if e1 > b0:
                extract_fn(b0, e1, buf[b0:e1])
            ev = torch.cuda.Event()
            ev.record(compute)
            self.cstream.wait_event(ev)
            self.comm.gather_rows(buf[b0:max(b0, e1)], self.slice_counts[k], self.slice_dst[k], root=self.dst)
        done = torch.cuda.Event()
        done.record(self.cstream)
        self.buf_free[b] = done
        self.last_buf = buf
        self.steps += 1

    def step(self, extract_fn: Callable[[int, int, object], None]):
        if self.transport == "nfx":
            return self._step_nfx(extract_fn)
        buf = self.bufs[self.steps & 1]
        works = []
        for k in range(len(self.bounds) - 1):
            b0, b1 = self.bounds[k], self.bounds[k + 1]
            e1 = min(b1, self.n_local)
            if e1 > b0:
                extract_fn(b0, e1, buf[b0:e1])
            send = buf[b0:b1] if self.wire == self.device else buf[b0:b1].to(self.wire)
            recv = [blk[b0:b1] for blk in self.blocks] if self.rank == self.dst else None
            works.append(self.dist.gather(send, recv, dst=self.dst, group=self.group, async_op=True))
        # the gathers of the previous step stayed in flight during this step's compute; their buffer is written again
        # in the NEXT step, so they are waited for here (stream-ordered on RCCL, host-side on gloo)
        for w in self.pending:
            w.wait()
        self.pending = works
        self.last_buf = buf
        self.steps += 1

    def finish(self):
        if self.transport == "nfx":
            self.cstream.synchronize()
            return
        for w in self.pending:
            w.wait()
        self.pending = []

    def _received(self, r):
        """On dst: rank r's rows of the last step as they arrived ([counts[r], F])."""
        if self.transport == "nfx":
            import torch
            parts = []
            for k, sc in enumerate(self.slice_counts):
                o = sum(sc[:r])
                parts.append(self.slice_dst[k][o:o + sc[r]])
            return torch.cat(parts, dim=0)
        return self.blocks[r][:self.counts[r]]

    def local_rows(self):
        """This rank's rows of the last step ([n_local, F])."""
        return self.last_buf[:self.n_local]

    def result(self):
        """On dst: the gathered [n_total, F] matrix in index order (a view when the shards are equal)."""
        if self.rank != self.dst:
            return None
        self.finish()
        import torch
        if self.transport == "nfx":
            return torch.cat([self._received(r) for r in range(self.world)], dim=0)
        if all(c == self.m for c in self.counts):
            return self.full[:self.n_total]
        return torch.cat([self.blocks[r][:self.counts[r]] for r in range(self.world)], dim=0)

    def verify_transport(self) -> Optional[List[bool]]:
        """Every rank checksums the rows it sent, dst checksums the rows it received: per-rank equality on dst."""
        import torch

        self.finish()
        mine = torch.tensor([row_checksum(self.local_rows())], dtype=torch.int64, device=self.wire)
        got = [torch.zeros(1, dtype=torch.int64, device=self.wire) for _ in range(self.world)] if self.rank == self.dst else None
        self.dist.gather(mine, got, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return [int(got[r].item()) == row_checksum(self._received(r)) for r in range(self.world)]

    def close(self):
        if self.transport == "nfx":
            self.finish()
            self.comm.close()
            self.cctx.set_stream(0)
            self.cctx.close()


# ---- NUMA placement of a rank ---------------------------------------------------------------------------------------
# Every rank of the streamed configuration (BASELINE configs[4]) pulls its slide -- 6.8 GB per step -- out of pinned host memory and
# writes 2 GB of rows back.  Pinned pages are placed where the pinning thread runs (first touch / local allocation), so a rank whose
# CPU threads run on the other socket streams across the socket interconnect, and eight such ranks share it.  Before its first GPU
# call a rank therefore binds itself to the CPUs of ITS GPU's NUMA node: `nfx.pinned_empty`, the registered slice of the shared row
# matrix and the slide are then first-touched locally.  (The reference picks `gpus[thread % n]` per rayon worker, utils.rs:211-224,
# and leaves placement to the OS.)
def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus: List[int] = []
    for part in (text or "").replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def visible_device_index(local_index: int, env=None) -> int:
    """The physical (KFD order) GPU behind HIP device `local_index`: ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    re-number the devices a process sees.  Lists of plain integers are followed (ROCR first, then HIP / CUDA on top of it); anything
    else (UUIDs) leaves the index as it is."""
    env = os.environ if env is None else env
    idx = local_index
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):   # HIP's list indexes into ROCR's
        val = env.get(name)
        if not val:
            continue
        try:
            ids = [int(x) for x in val.split(",") if x.strip() != ""]
        except ValueError:
            return idx
        if 0 <= idx < len(ids):
            idx = ids[idx]
    return idx


def gpu_numa_nodes(sysfs: str = "/sys") -> List[Optional[int]]:
    """NUMA node of every GPU in the order HIP enumerates them (KFD topology order): for each KFD node with SIMDs, its
    `drm_render_minor` names /sys/class/drm/renderD<minor>/device/numa_node.  -1 in sysfs (no affinity recorded) reads as None."""
    top = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((d for d in os.listdir(top) if d.isdigit()), key=int)
    except OSError:
        nodes = []
    out: List[Optional[int]] = []
    for d in nodes:
        props = {}
        for ln in (_read(os.path.join(top, d, "properties")) or "").splitlines():
            kv = ln.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue                                    # a CPU node of the topology
        minor = props.get("drm_render_minor")
        node = _read(os.path.join(sysfs, "class", "drm", f"renderD{minor}", "device", "numa_node")) if minor is not None else None
        try:
            out.append(int(node) if node is not None and int(node) >= 0 else None)
        except ValueError:
            out.append(None)
    if out:
        return out
    # no KFD topology (or an unreadable one): the DRM cards of vendor 0x1002 in card order
    drm = os.path.join(sysfs, "class", "drm")
    try:
        cards = sorted((c for c in os.listdir(drm) if c.startswith("card") and c[4:].isdigit()), key=lambda c: int(c[4:]))
    except OSError:
        cards = []
    for c in cards:
        if (_read(os.path.join(drm, c, "device", "vendor")) or "").lower() != "0x1002":
            continue
        node = _read(os.path.join(drm, c, "device", "numa_node"))
        try:
            out.append(int(node) if node is not None and int(node) >= 0 else None)
        except ValueError:
            out.append(None)
    return out


def bind_rank_to_gpu_numa(local_index: int, sysfs: str = "/sys", env=None, getaffinity=None, setaffinity=None) -> dict:
    """Binds the calling process to the CPUs of the NUMA node of HIP device `local_index` (intersected with the CPUs it is allowed
    to run on).  Call BEFORE the first GPU call and before any pinned allocation.  Touches nothing but sysfs and the scheduler.
    Returns what it found and did -- bench.py records it per rank as `config.numa`:
        {"gpu": physical index, "node": NUMA node or None, "cpus": CPUs bound to (count), "bound": bool, "why": text when not bound}"""
    getaffinity = getaffinity or (lambda: os.sched_getaffinity(0))
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    gpu = visible_device_index(local_index, env)
    nodes = gpu_numa_nodes(sysfs)
    if not (0 <= gpu < len(nodes)) and 0 <= local_index < len(nodes):
        # the visible-device list names GPUs this process cannot even see in sysfs (a container that hides the others' KFD nodes AND keeps the
        # host's numbering in the variable): the readable GPU nodes are then exactly the visible devices, in order
        gpu = local_index
    rec = {"gpu": gpu, "node": None, "cpus": None, "bound": False, "why": None}
    if not (0 <= gpu < len(nodes)):
        rec["why"] = f"sysfs lists {len(nodes)} GPU(s)"
        return rec
    node = nodes[gpu]
    rec["node"] = node
    if node is None:
        rec["why"] = "the GPU's numa_node is -1 (one node, or no affinity recorded)"
        return rec
    cpus = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) or "")
    allowed = set(getaffinity())
    mine = sorted(allowed.intersection(cpus))
    if not mine:
        rec["why"] = f"none of node {node}'s {len(cpus)} CPUs is in this process's affinity mask ({len(allowed)} CPUs)"
        return rec
    if set(mine) == allowed:
        rec.update(cpus=len(mine), bound=True, why="already confined to the node")
        return rec
    try:
        setaffinity(mine)
    except OSError as e:
        rec["why"] = f"sched_setaffinity refused: {e!r}"
        return rec
    rec.update(cpus=len(mine), bound=True)
    return rec


# ---- process launch -----------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_command(script: str, n_ranks: int, argv: Sequence[str], port: Optional[int] = None) -> List[str]:
    """The command line that runs `script argv` as n_ranks ranks of one node (the form the driver itself uses)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def launch_ranks(script: str, n_ranks: int, argv: Sequence[str]) -> int:
    """Starts `script` as n_ranks CHILD processes and relays their output; returns the launcher's exit code.

    Must be called before the calling process has touched the GPU (no HIP call, no torch.cuda.is_available()):
    the children are started with subprocess -- never exec'd over a process that holds a GPU context."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL peer access across processes on this host driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.run(rank_command(script, n_ranks, argv), env=env, stdout=subprocess.PIPE, text=True)
    # rank 0's one JSON line is the contract; whatever else reached the children's stdout goes to stderr
    json_lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in proc.stdout.splitlines():
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(json_lines) == 1:
        print(json_lines[0], flush=True)
        return 0
    print(f"launch_ranks: {n_ranks} ranks exited with code {proc.returncode}, {len(json_lines)} result lines", file=sys.stderr)
    return proc.returncode or 1
