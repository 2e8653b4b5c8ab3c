"""N>1 path on CPU: world-size-2 gloo through `distributed.ShardedExtraction` -- the class bench.py runs on the
GPUs.  The CPU oracle stands in for the HIP engine as the per-slice extractor (this test is about sharding, slicing,
double-buffered gathers and row order, not the kernels); the gathered matrix must equal the single-process result
row for row, including the batch-coupled mean_h."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "nuclei-feature-extraction_amd"


def _worker(rank, world, port, n, ps, batch, shard_batch, n_slices, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle as orc
    import _parity as par
    D = importlib.import_module(PKG + ".distributed")
    off, xy, pu8 = par.make_inputs(31, n, ps)
    sets = [0, 1]
    sx = D.ShardedExtraction(n, 30, shard_batch, n_slices=n_slices)

    def extract(b0, b1, out):   # local rows [b0, b1) of this rank's shard
        i0, i1 = sx.i0 + b0, sx.i0 + b1
        # the reference batch partition restarts at the shard's first nucleus (a rank sees only its shard)
        assert b0 % batch == 0 or shard_batch % batch != 0
        so = (off[i0:i1 + 1] - off[i0]).astype(np.uint32)
        rows, _ = orc.extract(sets, so, xy[off[i0]:off[i1]], pu8[i0:i1], ps, batch_size=batch)
        out.copy_(torch.from_numpy(rows))

    for _ in range(steps):      # several steps: the two row buffers alternate and gathers stay in flight across steps
        sx.step(extract)
    ok = sx.verify_transport()
    full = sx.result()
    if rank == 0:
        q.put((full.numpy().copy(), ok, sx.counts, sx.bounds))
    dist.barrier()
    dist.destroy_process_group()


def _run(n=70, ps=32, batch=16, shard_batch=16, n_slices=2, steps=3, world=2):
    D = importlib.import_module(PKG + ".distributed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D.free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ps, batch, shard_batch, n_slices, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, ok, counts, bounds = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    import _parity as par
    off, xy, pu8 = par.make_inputs(31, n, ps)
    want, _ = orc.extract([0, 1], off, xy, pu8, ps, batch_size=batch)
    return got, want, ok, counts, bounds


def test_sharded_gather_equals_single_process():
    got, want, ok, counts, bounds = _run()
    assert counts == [48, 22] and bounds == [0, 16, 48]      # ragged shards, two slices
    assert ok == [True, True]
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def test_equal_shards_single_slice():
    got, want, ok, counts, _ = _run(n=64, n_slices=1, steps=1)
    assert counts == [32, 32] and ok == [True, True]
    assert np.array_equal(got, want, equal_nan=True)


def test_eight_ranks_ragged_and_empty_shards():
    """The target world size (8 ranks, gloo on the CPU): 181 nuclei in batches of 16 are 12 batches, two per rank -- ranks 0..4 hold 32
    nuclei, rank 5 holds the ragged tail (16 + 5), ranks 6 and 7 hold NOTHING; every rank still takes part in every gather, over
    several steps (both row buffers in use), and the gathered matrix is the single-process one row for row."""
    got, want, ok, counts, bounds = _run(n=181, world=8, n_slices=2, steps=3)
    assert counts == [32, 32, 32, 32, 32, 21, 0, 0] and bounds == [0, 16, 32]
    assert ok == [True] * 8
    assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True)


def test_eight_ranks_fewer_batches_than_ranks():
    """n < 8 batches: 40 nuclei are three batches of 16, so ranks 3..7 have no shard at all (main.rs:146-148 with fewer chunks than workers)."""
    got, want, ok, counts, bounds = _run(n=40, world=8, n_slices=4, steps=2)
    assert counts == [16, 16, 8, 0, 0, 0, 0, 0] and bounds == [0, 16]       # one batch per rank: the four slices collapse to one
    assert ok == [True] * 8
    assert np.array_equal(got, want, equal_nan=True)


def test_eight_ranks_equal_shards():
    got, want, ok, counts, _ = _run(n=8 * 32, world=8, n_slices=2, steps=2)
    assert counts == [32] * 8 and ok == [True] * 8
    assert np.array_equal(got, want, equal_nan=True)


def test_shard_cut_inside_a_batch_changes_mean_h():
    """Why shard boundaries must be multiples of the reference batch: mean_h (column 12+6) couples a batch.
    Shards aligned to 20 nuclei cut the reference batches of 16."""
    got, want, ok, counts, _ = _run(shard_batch=20, n_slices=1, steps=1)
    assert counts == [40, 30] and ok == [True, True]
    col = 12 + 6
    assert not np.allclose(got[:, col], want[:, col], rtol=1e-6, atol=1e-6, equal_nan=True)
    other = [c for c in range(got.shape[1]) if c not in (col, 12 + 9)]
    assert np.array_equal(got[:, other], want[:, other], equal_nan=True)


def test_shard_ranges_cover_in_order():
    D = importlib.import_module(PKG + ".distributed")
    for n, B, R in [(1000, 100, 8), (1050, 100, 8), (70, 16, 2), (5, 100, 4), (0, 100, 2), (1_000_000, 100, 8)]:
        prev = 0
        for r in range(R):
            i0, i1 = D.shard_range(n, B, r, R)
            assert i0 == prev and i0 % B == 0 or i0 == n
            prev = i1
        assert prev == n
        assert sum(D.shard_counts(n, B, R)) == n
    for rows, B, k in [(125_000, 100, 4), (70, 16, 4), (48, 16, 8), (0, 100, 4), (1, 100, 4)]:
        b = D.slice_bounds(rows, B, k)
        assert b[0] == 0 and b[-1] == rows and all(x <= y for x, y in zip(b, b[1:]))
        assert all(x % B == 0 for x in b[:-1])


def test_row_checksum_sees_order_and_nan_payload():
    D = importlib.import_module(PKG + ".distributed")
    a = torch.tensor([[1.0, 2.0], [float("nan"), 4.0]])
    b = a.flip(0)
    assert D.row_checksum(a) == D.row_checksum(a.clone())
    assert D.row_checksum(a) != D.row_checksum(b)


def test_rank_command_is_the_drivers_form():
    D = importlib.import_module(PKG + ".distributed")
    cmd = D.rank_command("bench.py", 4, ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                       "--master-port", "29511", "bench.py", "--gpus", "4", "--steps", "3"]


def test_launch_ranks_relays_one_json_line(tmp_path):
    """The self-launch path of `python bench.py --gpus N` (no GPU needed: the child is a stub that reports its rank env)."""
    D = importlib.import_module(PKG + ".distributed")
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "print('noise from a library')\n"
        "if os.environ['RANK'] == '0':\n"
        "    print(json.dumps({'world': int(os.environ['WORLD_SIZE']), 'argv': sys.argv[1:]}))\n"
        "sys.exit(int(sys.argv[1]))\n")
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import importlib; D = importlib.import_module(%r + '.distributed'); "
            "sys.exit(D.launch_ranks(%r, 2, sys.argv[1:]))" % (ROOT, PKG, str(stub)))
    r = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip()) == {"world": 2, "argv": ["0"]}
    assert "noise from a library" in r.stderr and "noise" not in r.stdout
    r = subprocess.run([sys.executable, "-c", code, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


# ---- NUMA placement of a rank (distributed.bind_rank_to_gpu_numa): a fake sysfs tree, no GPU, no real scheduler call -------------------
def _fake_sysfs(tmp_path, gpu_nodes, node_cpus, cpu_topology_nodes=2, with_kfd=True):
    """A two-socket host as sysfs shows it: KFD topology nodes 0..cpu_topology_nodes-1 are the sockets (simd_count 0), the GPUs follow in
    HIP's enumeration order, GPU g is /sys/class/drm/renderD{128+g} and sits on NUMA node gpu_nodes[g]."""
    sysfs = tmp_path / "sys"
    if with_kfd:
        top = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
        for c in range(cpu_topology_nodes):
            (top / str(c)).mkdir(parents=True)
            (top / str(c) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        for g in range(len(gpu_nodes)):
            d = top / str(cpu_topology_nodes + g)
            d.mkdir(parents=True)
            (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
    for g, node in enumerate(gpu_nodes):
        for name in (f"renderD{128 + g}", f"card{g}"):
            dev = sysfs / "class" / "drm" / name / "device"
            dev.mkdir(parents=True)
            (dev / "numa_node").write_text(f"{node}\n")
            (dev / "vendor").write_text("0x1002\n")
    for node, cpus in node_cpus.items():
        d = sysfs / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    return str(sysfs)


def test_a_rank_binds_to_the_cpus_of_its_gpus_numa_node(tmp_path):
    D = importlib.import_module(PKG + ".distributed")
    sysfs = _fake_sysfs(tmp_path, gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], node_cpus={0: "0-63,128-191", 1: "64-127,192-255"})
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and D.parse_cpulist("") == []
    assert D.gpu_numa_nodes(sysfs) == [0, 0, 0, 0, 1, 1, 1, 1]
    for local_rank, node, first_cpu in ((0, 0, 0), (3, 0, 0), (4, 1, 64), (7, 1, 64)):
        calls = []
        rec = D.bind_rank_to_gpu_numa(local_rank, sysfs=sysfs, env={}, getaffinity=lambda: set(range(256)), setaffinity=calls.append)
        assert rec["bound"] and rec["node"] == node and rec["gpu"] == local_rank and rec["cpus"] == 128 and rec["why"] is None
        assert len(calls) == 1 and calls[0][0] == first_cpu and len(calls[0]) == 128
        assert set(calls[0]) == set(D.parse_cpulist({0: "0-63,128-191", 1: "64-127,192-255"}[node]))
    # a cgroup that grants only some CPUs: the intersection is what the rank is bound to; none of the node's CPUs allowed -> left alone, and said
    calls = []
    rec = D.bind_rank_to_gpu_numa(5, sysfs=sysfs, env={}, getaffinity=lambda: set(range(60, 80)), setaffinity=calls.append)
    assert rec["bound"] and rec["cpus"] == 16 and calls == [list(range(64, 80))]
    calls = []
    rec = D.bind_rank_to_gpu_numa(5, sysfs=sysfs, env={}, getaffinity=lambda: set(range(0, 16)), setaffinity=calls.append)
    assert not rec["bound"] and calls == [] and "affinity mask" in rec["why"]
    # already confined to the node: nothing to do
    calls = []
    rec = D.bind_rank_to_gpu_numa(0, sysfs=sysfs, env={}, getaffinity=lambda: set(range(0, 8)), setaffinity=calls.append)
    assert rec["bound"] and calls == [] and rec["cpus"] == 8
    # the scheduler refuses: reported, not raised
    def refuse(cpus):
        raise PermissionError(1, "Operation not permitted")
    rec = D.bind_rank_to_gpu_numa(0, sysfs=sysfs, env={}, getaffinity=lambda: set(range(256)), setaffinity=refuse)
    assert not rec["bound"] and "refused" in rec["why"]


def test_numa_binding_follows_visible_devices_and_survives_missing_sysfs(tmp_path):
    D = importlib.import_module(PKG + ".distributed")
    sysfs = _fake_sysfs(tmp_path, gpu_nodes=[0, 0, 1, 1], node_cpus={0: "0-7", 1: "8-15"})
    # ROCR_VISIBLE_DEVICES=2,3: HIP device 0 is physical GPU 2 (node 1); HIP_VISIBLE_DEVICES indexes into ROCR's list
    assert D.visible_device_index(0, {"ROCR_VISIBLE_DEVICES": "2,3"}) == 2
    assert D.visible_device_index(1, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "1,0"}) == 2
    assert D.visible_device_index(1, {"HIP_VISIBLE_DEVICES": "GPU-deadbeef"}) == 1
    calls = []
    rec = D.bind_rank_to_gpu_numa(0, sysfs=sysfs, env={"ROCR_VISIBLE_DEVICES": "2,3"}, getaffinity=lambda: set(range(16)), setaffinity=calls.append)
    assert rec["gpu"] == 2 and rec["node"] == 1 and calls == [list(range(8, 16))]
    # a container that hides the other GPUs' KFD nodes but keeps the host's numbering in the variable: the one readable GPU is device 0
    hidden = _fake_sysfs(tmp_path / "hidden", gpu_nodes=[1], node_cpus={0: "0-7", 1: "8-15"})
    calls = []
    rec = D.bind_rank_to_gpu_numa(0, sysfs=hidden, env={"ROCR_VISIBLE_DEVICES": "5"}, getaffinity=lambda: set(range(16)), setaffinity=calls.append)
    assert rec["bound"] and rec["gpu"] == 0 and rec["node"] == 1 and calls == [list(range(8, 16))]
    # numa_node -1 (a one-node host, or a VM that records no affinity): nothing bound, reason given
    one = _fake_sysfs(tmp_path / "one", gpu_nodes=[-1], node_cpus={0: "0-15"})
    rec = D.bind_rank_to_gpu_numa(0, sysfs=one, env={}, getaffinity=lambda: set(range(16)), setaffinity=calls.append)
    assert not rec["bound"] and rec["node"] is None and "-1" in rec["why"]
    # no KFD topology: the DRM cards of vendor 0x1002, in card order
    drm = _fake_sysfs(tmp_path / "drm", gpu_nodes=[1, 0], node_cpus={0: "0-7", 1: "8-15"}, with_kfd=False)
    assert D.gpu_numa_nodes(drm) == [1, 0]
    # no sysfs at all (this container): nothing bound, nothing raised
    rec = D.bind_rank_to_gpu_numa(0, sysfs=str(tmp_path / "nowhere"), env={}, getaffinity=lambda: set(range(4)), setaffinity=calls.append)
    assert not rec["bound"] and "0 GPU(s)" in rec["why"]
    # and against the real /sys of wherever this runs, with the real scheduler call stubbed out: a dict, never an exception
    rec = D.bind_rank_to_gpu_numa(0, setaffinity=lambda cpus: None)
    assert set(rec) == {"gpu", "node", "cpus", "bound", "why"}
