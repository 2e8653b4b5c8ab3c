"""Worker of tests/test_comm_gpu.py::test_two_ranks_native_transport (one process per GPU, started by torch.distributed.run):
ShardedExtraction(transport="nfx") -- nfx_gather_rows over RCCL below any framework -- with RAGGED shards and an EMPTY shard.
Rank r's rows are a pure function of the nucleus index, so the root can check the gathered matrix element for element."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("gloo")          # control plane only (the 128-byte id, checksums): the rows travel through nfx_gather_rows
    nfxd = importlib.import_module("nuclei-feature-extraction_amd.distributed")
    F, B = 37, 10
    ok = True
    for n_total in (10 * B * world + 3 * B + 7,   # ragged: the last rank's shard is shorter and ends inside a batch
                    B):                           # one batch in all: every rank but 0 has an EMPTY shard
        sh = nfxd.ShardedExtraction(n_total, F, B, n_slices=3, device=dev, dst=0, transport="nfx")

        def fill(b0, b1, out, i0=sh.i0):
            idx = torch.arange(i0 + b0, i0 + b1, device=dev, dtype=torch.float32)
            out.copy_(idx[:, None] * 1000.0 + torch.arange(F, device=dev, dtype=torch.float32)[None, :])

        for _ in range(3):                        # several steps: the two row buffers alternate, gathers overlap the next step
            sh.step(fill)
        checks = sh.verify_transport()
        res = sh.result()
        if rank == 0:
            want = torch.arange(n_total, device=dev, dtype=torch.float32)[:, None] * 1000.0 + torch.arange(F, device=dev, dtype=torch.float32)[None, :]
            ok = ok and all(checks) and res.shape == want.shape and bool(torch.equal(res, want))
        sh.close()
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    if rank == 0:
        print("TWO_RANK_NFX_OK" if ok else "TWO_RANK_NFX_MISMATCH", flush=True)
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
