"""Write `proof.bin` (the reference's ExecutionProof bytes: postcard of Proof<F>) and `instance.bin` (what its verifier needs:
bytecode + public input) for the golden prove_execution instance of tests/golden/vectors_r01.json — the inputs of the
external parity pin in rust_shim/src/lib.rs (`cargo test` inside the reference's workspace feeds them to verify_execution).
  python tools/write_proof.py [out_dir] [--cpu] [--golden]
      default: the witness of the golden instance proven at the reference's default_whir_config (124-bit, 16 grinding bits) —
               what the unmodified reference verifier expects;
      --golden: the golden instance's own reduced parameters (50-bit), checked against tests/golden/vectors_r02.json;
      --cpu:    take the proof words from the CPU oracle instead of the GPU (same bytes — tests/test_golden.py).
instance.bin: u32 LE words  [log_bytecode, ending_pc, n_public_input, bytecode_hash x 8, public_input.., bytecode rows x 16..]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import leanmultisig_amd as lm  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests import synth_witness  # noqa: E402

out_dir = next((a for a in sys.argv[1:] if not a.startswith("--")), ROOT)
if "--xmss" in sys.argv:
    # Second fixture (round 4): the REAL-signature path.  The hand-assembled aggregation program (smallest power-of-two bytecode)
    # verifies 40 XMSS signatures; lmh_prove_execution_vm — leanVM run with the parallel batch on the device, trace build, proof at the
    # reference's default_whir_config — produces the proof.  rust_shim: reference_verifier_accepts_the_hip_proof_of_real_signatures.
    from leanmultisig_amd import vm
    from leanmultisig_amd.programs import xmss_aggregate as xa
    ctx = lm.Context(0)
    bc = xa.build_program()
    pi, wit, _ = xa.build_witness(bc, 40, np.random.default_rng(2026), xmss=xa.Xmss(compress=lambda x: ctx.poseidon16(x, compress=True)))
    lb = lm.WhirBuilder.default(1)
    pr = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, pr, bc, pi, wit, lb)
    data = pr.proof_bytes()
    ex = vm.execute(bc, pi, wit, ctx=ctx)
    assert ex.on_device
    w = dict(log_bytecode=bc.log_size, ending_pc=bc.ending_pc, public_input=np.asarray(pi, dtype=np.uint32), bytecode_hash=bc.hash(), bytecode=bc.multilinear)
    inst = np.concatenate([np.array([w["log_bytecode"], w["ending_pc"], w["public_input"].size], dtype=np.uint32), w["bytecode_hash"],
                           w["public_input"], w["bytecode"].reshape(-1)]).astype("<u4")
    os.makedirs(out_dir, exist_ok=True)
    open(os.path.join(out_dir, "proof.bin"), "wb").write(data)
    # instance.zlib: CANONICAL words (small integers compress; the bytecode table is 2^17 rows, of which ~98 k are the unrolled blocks of
    # the public-key hash): [log_bytecode, ending_pc, n_public_input, bytecode_hash x 8, public_input.., instruction rows x 12 columns..]
    import zlib
    from leanmultisig_amd.vm import from_monty
    canon = np.concatenate([inst[:3], from_monty(inst[3:11 + w["public_input"].size]).astype(np.uint32),
                            from_monty(w["bytecode"][:, :12]).astype(np.uint32).reshape(-1)]).astype("<u4")
    z = zlib.compress(canon.tobytes(), 9)
    open(os.path.join(out_dir, "instance.zlib"), "wb").write(z)
    ok, err = lm.verify_execution(w, data, None)  # the library's verifier, default_whir_config read off the proof as the reference does
    assert ok, err
    print(f"wrote {out_dir}/proof.bin ({len(data)} bytes, sha256 {hashlib.sha256(data).hexdigest()[:16]}..) and instance.zlib ({len(z)} bytes for {inst.size} words): "
          f"{ex.n_cycles} cycles, {ex.n_poseidon_calls} Poseidon calls, bytecode 2^{bc.log_size}")
    sys.exit(0)
v = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors_r01.json")))["prove_execution"]
orc = ob.load()
w = synth_witness.build(orc, np.random.default_rng(v["seed"]), n_calls=v["n_calls"])
golden = "--golden" in sys.argv
b = np.array(v["builder"], dtype=np.uint32) if golden else ob.whir_builder(log_inv_rate=1)
lb = lm.WhirBuilder.default(int(b[0]), security_level=int(b[6]), pow_bits=int(b[7]))
n_vars = synth_witness.stacked_n_vars(w)
cfg = lm.WhirConfig.new(lb, n_vars)
if "--cpu" in sys.argv:
    raw = ob.prove_execution(orc, w, synth_witness.header(w), b)
    c = cfg.to_dict()
    pr = lm.Prover.from_raw(raw, [r["num_queries"] for r in c["rounds"]] + [c["final_queries"]])
else:
    ctx = lm.Context(0)
    tr, keep = lm.make_execution_trace(ctx, w)
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, cfg)
data = pr.proof_bytes()
sha = hashlib.sha256(data).hexdigest()
if golden:
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors_r02.json")))["proof_bytes"]
    assert sha == gold["postcard_sha256"], "proof bytes differ from tests/golden/vectors_r02.json"
ok, err = lm.verify_execution(w, data, None if not golden else lb)  # None: default_whir_config read off the proof, as the reference does
assert ok, err
open(os.path.join(out_dir, "proof.bin"), "wb").write(data)
inst = np.concatenate([np.array([w["log_bytecode"], w["ending_pc"], w["public_input"].size], dtype=np.uint32), w["bytecode_hash"],
                       w["public_input"], w["bytecode"].reshape(-1)]).astype("<u4")
open(os.path.join(out_dir, "instance.bin"), "wb").write(inst.tobytes())
print(f"wrote {out_dir}/proof.bin ({len(data)} bytes, sha256 {sha[:16]}.., {int(b[6])}-bit / {int(b[7])} grinding bits) and instance.bin ({inst.size} words)")
