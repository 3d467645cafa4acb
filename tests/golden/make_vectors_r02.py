"""Generate tests/golden/vectors_r02.json: the reference's proof BYTES (postcard of Proof<F>, transcript.rs:33-36) of the
golden prove_execution instance of vectors_r01.json, produced by the CPU oracle (proof words, pruning) and the independent
Python restatement of the byte format (tests/wire_py.py).  The library's serialiser and the device prover must reproduce
them (tests/test_wire_format.py, tests/test_golden.py).  No Rust toolchain exists in this image, so — like vectors_r01 —
nothing here comes from running the reference itself.
usage: python tests/golden/make_vectors_r02.py   (from the repo root; needs oracle/liblm_oracle.so)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests import oracle_binding as ob  # noqa: E402
from tests import synth_witness, wire_py  # noqa: E402


def vectors(orc, sizes):
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors_r01.json")))["prove_execution"]
    w = synth_witness.build(orc, np.random.default_rng(v["seed"]), n_calls=v["n_calls"])
    raw = ob.prove_execution(orc, w, synth_witness.header(w), np.array(v["builder"], dtype=np.uint32))
    pruned = ob.prune_proof(orc, raw, sizes)
    data = wire_py.postcard_proof(pruned)
    return {"proof_bytes": {"instance": "vectors_r01.json: prove_execution", "batch_sizes": [int(s) for s in sizes],
                            "postcard_len": len(data), "postcard_sha256": hashlib.sha256(data).hexdigest(),
                            "proof_size_fe": int(ob.pruned_size_fe(orc, pruned))}}


if __name__ == "__main__":
    orc = ob.load()
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors_r01.json")))["prove_execution"]
    w = synth_witness.build(orc, np.random.default_rng(v["seed"]), n_calls=v["n_calls"])
    cfg = ob.whir_config(orc, np.array(v["builder"], dtype=np.uint32), synth_witness.stacked_n_vars(w))
    sizes = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    out = os.path.join(ROOT, "tests", "golden", "vectors_r02.json")
    json.dump(vectors(orc, sizes), open(out, "w"), indent=1)
    print("wrote", out)
