"""CPU test of the N > 1 path: world_size = 2 over gloo — signer partition and the only collective of the sharded path
(all-gather of the per-rank commitment root + pruned proof after every step, bench.py:exchange_step)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["LM_ROOT"])
import numpy as np, torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ranges = bench.signer_ranges(2 * 1550 + 1, world)
assert ranges[0][0] == 0 and ranges[-1][1] == 2 * 1550 + 1 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
root = np.arange(8, dtype=np.uint32) + 100 * rank + 0x7E000000      # words near p: the int32 transport must be lossless
proof = (np.arange(5000 + 37 * rank, dtype=np.uint32) * 2654435761 % 0x7F000001).astype(np.uint32)  # ranks send different lengths
allg = bench.exchange_step(root, proof, torch.device("cpu")).cpu().numpy()
assert allg.shape[0] == world
for r in range(world):
    assert list(allg[r, :8]) == list(np.arange(8) + 100 * r + 0x7E000000), allg[r, :8]
    n = int(allg[r, 8])
    assert n == 5000 + 37 * r
    want = (np.arange(n, dtype=np.uint32) * 2654435761 % 0x7F000001).astype(np.uint32)
    assert np.array_equal(allg[r, 9:9 + n].astype(np.uint32), want) and not allg[r, 9 + n:].any()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_root_allgather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2
