"""Throughput of C concurrent provers (one lm_ctx / stream / host thread each) on one GPU."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMP_NUM_THREADS"] = "1"


def main():
    import numpy as np
    import bench
    import leanmultisig_amd as lm
    CMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = 4
    ctxs, ws = [], []
    for c in range(CMAX):
        ctx = lm.Context(0)
        ctxs.append(ctx)
        ws.append(bench.build_vm_workload(ctx, np.random.default_rng(1000 + c), bench.N_SIGS, 1, False))
        bench.run_step(ctx, lm, ws[-1])
    for C in range(1, CMAX + 1):
        bar = threading.Barrier(C + 1)
        def worker(i):
            bar.wait()
            for _ in range(steps):
                bench.run_step(ctxs[i], lm, ws[i])
            ctxs[i].sync()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(C)]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(f"C={C}: {1e3 * dt / steps:.2f} ms per round of {C} proofs -> {1550 * C * steps / dt:.0f} sigs/s")


if __name__ == "__main__":
    main()
