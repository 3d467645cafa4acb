"""Race detector for the latency machinery (fence-free publishes, side streams, staging ring, pooled memory): C provers on one
GPU prove the SAME leaf over and over from C host threads; proofs are deterministic, so every proof of a prover must equal
its first one word for word, and EVERY proof is checked by lmh_verify_execution (a reordered publish would corrupt the
transcript: the proof is rejected).  usage: python tools/stress_inflight.py [provers] [proofs each] [scale_log] [whole]
whole = 1: every proof is the WHOLE function (lmh_prove_execution_vm: VM run with the parallel batch on the device, trace, proof).
hog = a command started (as a FOREIGN process: it knows nothing of this library) once the workloads are built and stopped when the
proofs are done, e.g. "tools/ubench/hog 248 600 1024 1": 248 workgroups that fill a compute unit each, 8 CUs left to the provers — a
16-workgroup resident GKR tail can then never be resident as a whole and must fail soft (lm_soft_fallbacks), not fail the proof."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench  # noqa: E402
import leanmultisig_amd as lm  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
WHOLE = len(sys.argv) > 4 and sys.argv[4] == "1"
HOG = sys.argv[5] if len(sys.argv) > 5 else ""
ctxs = [lm.Context(0) for _ in range(C)]
ws = [bench.build_vm_workload(ctxs[c], np.random.default_rng(900 + c), max(2, bench.N_SIGS >> SCALE), 1, False, log_bytecode=19 if SCALE == 0 else None)
      for c in range(C)]
first = []
for c in range(C):
    pr = bench.run_step(ctxs[c], lm, ws[c], whole_node=False)
    ok, err = lm.verify_execution(ws[c]["w"], pr.proof_bytes(compressed=True), ws[c]["lm_builder"], compressed=True)
    assert ok, err
    first.append(pr.proof().copy())
bad = []
start = threading.Barrier(C)


def worker(c):
    try:
        ctxs[c]._check(ctxs[c].lib.lm_bind_thread(ctxs[c].h))   # the HIP device is per host thread
        start.wait()
        for i in range(N):
            pr = bench.run_step(ctxs[c], lm, ws[c], whole_node=WHOLE)
            p = pr.proof()
            ok, err = lm.verify_execution(ws[c]["w"], pr, ws[c]["lm_builder"])   # every proof goes through lmh_verify_execution
            if not ok:
                bad.append((c, i, "rejected: " + err))
            if p.size != first[c].size or not np.array_equal(p, first[c]):
                bad.append((c, i, int(np.argmax(p[:min(p.size, first[c].size)] != first[c][:min(p.size, first[c].size)]))))
    except Exception as e:  # noqa: BLE001
        bad.append((c, -1, repr(e)))


hog = None
if HOG:
    import shlex
    import subprocess
    hog = subprocess.Popen(shlex.split(HOG), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(1.0)   # its first kernel is resident
t0 = time.time()
th = [threading.Thread(target=worker, args=(c,)) for c in range(C)]
[t.start() for t in th]
[t.join() for t in th]
if hog is not None:
    hog.terminate()   # (this exact process)
    hog.wait()
print(("under `" + HOG + "`: " if HOG else "") + f"{C} provers x {N} proofs in {time.time() - t0:.1f} s: {len(bad)} proofs differ from the prover's first proof {bad[:5]}; "
      f"GKR layers re-run without resident kernels (lm_soft_fallbacks): {sum(c.soft_fallbacks() for c in ctxs)}")
sys.exit(1 if bad else 0)
