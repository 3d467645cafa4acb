"""GPU: the N > 1 bench path end to end on a single-GPU box — two ranks (gloo, both on GPU 0), one proof per rank and step,
the all-gather of every leaf's commitment root and pruned proof, max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # two processes share the chip: the resident GKR tail workgroups of both are counted per DEVICE (csrc/lm_gkr.hip: a shared-memory
    # counter keyed by the PCI bus id), no per-process share has to be configured
    env = dict(os.environ, LM_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    env.pop("LM_GKR_TAIL_MAX_WORKGROUPS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scale-log", "5", "--dist-backend", "gloo", "--no-cpu-baseline", "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["proof_verified_by_oracle"] is True and "inflight" not in j  # the in-flight side measurement is N = 1 only
    # the headline is the whole node at every N: VM run (parallel batch on the device) + trace + proof per step and rank
    assert j["whole_node"]["value"] == j["value"] and j["whole_node"]["vm_run_ms"] > 0 and j["hot_path"]["ms_per_step"] < j["ms_per_step"]
    assert j["hot_path"]["proof_equals_whole_node_proof"] is True
    assert "Witness generation: Executing bytecode" in j["stages_ms"] and "batched AIR sumcheck" in j["stages_ms"]
    # round 5: the step is aggregate_type_1 whole (input assembly included), and the line says where the VM's batch ran
    assert j["stages_ms"]["aggregate_type_1: inputs"] > 0 and j["vm_on_device"] is True and j["host_batches"] == 0
    assert j["exchanges"]["per_step"] > 10 and j["node_stats"]["n_xmss"] == j["config"]["per_gpu_signatures"]
    # the host side of every rank: own work per step, CPU time, jitter between the ranks
    rk = j["ranks"]
    assert len(rk["per_rank"]) == 2 and all(0 < r["host_busy_ms_per_step"] < r["ms_per_step"] and r["cpu_ms_per_step"] > 0 for r in rk["per_rank"])
    assert rk["step_jitter_ms"]["max"] >= 0
