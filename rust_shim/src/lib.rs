//! Rust side of the C ABI of `include/leanmultisig.h` / `include/leanmultisig_host.h` (UNCOMPILED here, see Cargo.toml).
//!
//! Two uses:
//!  * `prove_execution_hip`: the coarse drop-in for `lean_prover::prove_execution` after witness generation — the trace is
//!    uploaded once, the whole proof is produced on the MI355X and comes back as the reference's own `ExecutionProof` bytes;
//!  * the `#[test]` below: feed a proof file written by this repository (`python tools/write_proof.py proof.bin`) to the
//!    reference's `verify_execution` — an external parity pin that needs no GPU on the Rust side.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct lm_ctx(c_void);
#[repr(C)]
pub struct lmh_prover(c_void);

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lm_whir_round {
    pub query_pow_bits: u32,
    pub folding_pow_bits: u32,
    pub num_queries: u32,
    pub ood_samples: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lm_whir_config {
    pub num_variables: u32,
    pub starting_log_inv_rate: u32,
    pub folding_factor_first: u32,
    pub folding_factor_subsequent: u32,
    pub rs_domain_initial_reduction_factor: u32,
    pub commitment_ood_samples: u32,
    pub starting_folding_pow_bits: u32,
    pub n_rounds: u32,
    pub final_queries: u32,
    pub final_query_pow_bits: u32,
    pub final_sumcheck_rounds: u32,
    pub rounds: [lm_whir_round; 8],
}
#[repr(C)]
pub struct lm_vm_table {
    pub log_rows: u32,
    pub non_padded_n_rows: u32, // TableTrace::non_padded_n_rows (0 = unknown: every row is summed)
    pub d_cols: *const *const u32,
}
#[repr(C)]
pub struct lm_execution_trace {
    pub log_inv_rate: u32,
    pub log_memory: u32,
    pub log_bytecode: u32,
    pub ending_pc: u32,
    pub public_memory_size: u32,
    pub n_public_input: u32,
    pub public_input: *const u32,
    pub bytecode_hash: *const u32,
    pub d_bytecode: *const u32,
    pub d_bytecode_acc: *const u32,
    pub d_memory: *const u32,
    pub d_memory_acc: *const u32,
    pub tables: [lm_vm_table; 3],
    pub d_stacked: *mut u32, // optional: the trace already in its committed layout (null = absent)
}
#[repr(C)]
pub struct lmh_bytecode(c_void);
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lm_vm_hint {
    pub pc: u32,
    pub kind: u32, // LM_VM_HINT_*
    pub args: [u32; 4],
    pub mode: [u8; 4], // LM_VM_ARG_*: 0 constant, 1 m[fp + x], 2 fp + x
}
#[repr(C)]
pub struct lm_vm_witness {
    pub preamble_memory_len: u32,
    pub n_names: u32,
    pub name_entry_begin: *const u64,
    pub entry_offset: *const u64,
    pub data: *const u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lm_whir_builder {
    pub starting_log_inv_rate: u32,
    pub max_num_variables_to_send_coeffs: u32,
    pub rs_domain_initial_reduction_factor: u32,
    pub folding_factor_first: u32,
    pub folding_factor_subsequent: u32,
    pub soundness_type: u32,
    pub security_level: u32,
    pub pow_bits: u32,
}

unsafe extern "C" {
    pub fn lmh_bytecode_new(instructions_multilinear: *const u32, log_size: u32, n_instructions: u64, ending_pc: u32, starting_frame_memory: u32,
                            hints: *const lm_vm_hint, n_hints: u64, n_hint_names: u32) -> *mut lmh_bytecode;
    pub fn lmh_bytecode_free(bc: *mut lmh_bytecode);
    pub fn lmh_default_whir_builder(starting_log_inv_rate: u32, prox_gaps_conjecture: c_int, out: *mut lm_whir_builder);
    pub fn lmh_prove_execution_vm(ctx: *mut lm_ctx, p: *mut lmh_prover, bc: *const lmh_bytecode, public_input: *const u32, n_public_input: u32,
                                  witness: *const lm_vm_witness, builder: *const lm_whir_builder, n_threads: u32, times_ms: *mut f64) -> c_int;
    pub fn lm_ctx_create(device: c_int, out: *mut *mut lm_ctx) -> c_int;
    pub fn lm_ctx_destroy(ctx: *mut lm_ctx);
    pub fn lm_last_error() -> *const c_char;
    /// GKR layers this context re-ran with one launch per exchange because a resident kernel never got its wave slots on a shared
    /// device: a scheduling event for NodeStats, never a `ProverError` (the proof is unchanged)
    pub fn lm_soft_fallbacks(ctx: *const lm_ctx) -> u32;
    pub fn lm_malloc(ctx: *mut lm_ctx, n_words: u64, d_out: *mut *mut u32) -> c_int;
    pub fn lm_free(ctx: *mut lm_ctx, d_ptr: *mut u32) -> c_int;
    pub fn lm_upload(ctx: *mut lm_ctx, d_dst: *mut u32, src: *const u32, n_words: u64) -> c_int;
    pub fn lmh_prover_new() -> *mut lmh_prover;
    pub fn lmh_prover_free(p: *mut lmh_prover);
    pub fn lmh_stacked_n_vars(trace: *const lm_execution_trace) -> u32;
    pub fn lmh_prove_execution(ctx: *mut lm_ctx, p: *mut lmh_prover, trace: *const lm_execution_trace, cfg: *const lm_whir_config) -> c_int;
    pub fn lmh_proof_postcard_size(p: *const lmh_prover) -> u64;
    pub fn lmh_proof_postcard(p: *const lmh_prover, out: *mut u8);
}

/// `WhirConfig<EF>` (crates/whir/src/config.rs:118-134) -> the integers the library takes (never re-derived on the far side).
pub fn whir_config_ints(c: &backend::WhirConfig<backend::EF>) -> lm_whir_config {
    let mut o = lm_whir_config {
        num_variables: c.num_variables as u32,
        starting_log_inv_rate: c.starting_log_inv_rate as u32,
        folding_factor_first: c.folding_factor.at_round(0) as u32,
        folding_factor_subsequent: c.folding_factor.at_round(1) as u32,
        rs_domain_initial_reduction_factor: c.rs_domain_initial_reduction_factor as u32,
        commitment_ood_samples: c.commitment_ood_samples as u32,
        starting_folding_pow_bits: c.starting_folding_pow_bits as u32,
        n_rounds: c.round_parameters.len() as u32,
        final_queries: c.final_queries as u32,
        final_query_pow_bits: c.final_query_pow_bits as u32,
        final_sumcheck_rounds: c.final_sumcheck_rounds as u32,
        ..Default::default()
    };
    for (i, r) in c.round_parameters.iter().enumerate() {
        o.rounds[i] = lm_whir_round {
            query_pow_bits: r.query_pow_bits as u32,
            folding_pow_bits: r.folding_pow_bits as u32,
            num_queries: r.num_queries as u32,
            ood_samples: r.ood_samples as u32,
        };
    }
    o
}

/// `Bytecode` -> the library's object: `instructions_multilinear` as it is (the runner decodes its instructions from it) and the
/// hints attached to each pc, flattened (crates/lean_vm/src/isa/hint.rs:17-81).  `names` collects the `HintWitness` names in
/// first-use order: index = the id the witness streams are passed under.  Print / LocationReport / Label / Panic have no effect
/// on the execution result and are dropped.
pub fn bytecode_to_hip(bytecode: &lean_vm::Bytecode, names: &mut Vec<String>) -> *mut lmh_bytecode {
    use lean_vm::{CustomHint, Hint, HintWitnessDestination, MemOrConstant, MemOrFpOrConstant};
    fn moc(a: &MemOrConstant) -> (u32, u8) {
        match a {
            MemOrConstant::Constant(c) => (c.as_canonical_u32(), 0),
            MemOrConstant::MemoryAfterFp { offset } => (*offset as u32, 1),
        }
    }
    fn mfc(a: &MemOrFpOrConstant) -> (u32, u8) {
        match a {
            MemOrFpOrConstant::Constant(c) => (c.as_canonical_u32(), 0),
            MemOrFpOrConstant::MemoryAfterFp { offset } => (*offset as u32, 1),
            MemOrFpOrConstant::FpRelative { offset } => (*offset as u32, 2),
        }
    }
    let mut flat: Vec<lm_vm_hint> = Vec::new();
    for (pc, entry) in bytecode.code.iter().enumerate() {
        for h in entry.hints.iter() {
            let mut o = lm_vm_hint { pc: pc as u32, ..Default::default() };
            let mut set = |k: usize, (v, m): (u32, u8)| {
                o.args[k] = v;
                o.mode[k] = m;
            };
            match h {
                Hint::Inverse { arg, res_offset } => { o.kind = 1; set(0, moc(arg)); set(1, (*res_offset as u32, 0)); }
                Hint::RequestMemory { offset, size } => { o.kind = 2; set(0, (*offset as u32, 0)); set(1, moc(size)); }
                Hint::DerefHint { offset_src, offset_target } => { o.kind = 3; set(0, (*offset_src as u32, 0)); set(1, (*offset_target as u32, 0)); }
                Hint::Custom(c, args) => {
                    o.kind = match c { CustomHint::DecomposeBitsXMSS => 4, CustomHint::DecomposeBitsMerkleWhir => 5, CustomHint::DecomposeBits => 6,
                                       CustomHint::LessThan => 7, CustomHint::Log2Ceil => 8 };
                    for (k, a) in args.iter().enumerate() { set(k, mfc(a)); }
                }
                Hint::HintWitness { name, destination } => {
                    let id = names.iter().position(|n| n == name).unwrap_or_else(|| { names.push(name.clone()); names.len() - 1 }) as u32;
                    match destination {
                        HintWitnessDestination::Inline { offset } => { o.kind = 9; set(0, (id, 0)); set(1, (*offset as u32, 0)); }
                        HintWitnessDestination::Indirect { ptr_offset } => { o.kind = 10; set(0, (id, 0)); set(1, (*ptr_offset as u32, 0)); }
                    }
                }
                Hint::ParallelBatchStart { n_args, end_value } => { o.kind = 11; set(0, (*n_args as u32, 0)); set(1, moc(end_value)); }
                Hint::DebugAssert { expr, preceds_runtime_inequality, .. } => {
                    o.kind = 12; set(0, moc(&expr.left)); set(1, moc(&expr.right)); set(2, (expr.kind as u32, 0)); set(3, (*preceds_runtime_inequality as u32, 0));
                }
                Hint::Print { .. } | Hint::LocationReport { .. } | Hint::Label { .. } | Hint::Panic { .. } => continue,
            }
            flat.push(o);
        }
    }
    unsafe {
        lmh_bytecode_new(bytecode.instructions_multilinear.as_ptr() as *const u32, bytecode.log_size() as u32, bytecode.code.len() as u64,
                         bytecode.ending_pc as u32, bytecode.starting_frame_memory as u32, flat.as_ptr(), flat.len() as u64, names.len() as u32)
    }
}

/// The whole of `lean_prover::prove_execution` on the library: VM run (host thread pool), execution trace (device), proof.
pub fn prove_execution_vm_hip(ctx: *mut lm_ctx, bc: *const lmh_bytecode, names: &[String], public_input: &[backend::F],
                              witness: &lean_vm::ExecutionWitness, log_inv_rate: usize) -> Option<backend::Proof<backend::F>> {
    let (mut begin, mut offs, mut data) = (vec![0u64], vec![0u64], Vec::<u32>::new());
    for name in names {
        for entry in witness.hints.get(name).map(|v| v.as_slice()).unwrap_or(&[]) {
            data.extend(entry.iter().map(|f| f.to_monty_u32())); // the in-memory word of MontyField31
            offs.push(data.len() as u64);
        }
        begin.push(offs.len() as u64 - 1);
    }
    let w = lm_vm_witness { preamble_memory_len: witness.preamble_memory_len as u32, n_names: names.len() as u32,
                            name_entry_begin: begin.as_ptr(), entry_offset: offs.as_ptr(), data: data.as_ptr() };
    unsafe {
        let mut b = lm_whir_builder::default();
        lmh_default_whir_builder(log_inv_rate as u32, cfg!(feature = "prox-gaps-conjecture") as c_int, &mut b);
        let p = lmh_prover_new();
        let rc = lmh_prove_execution_vm(ctx, p, bc, public_input.as_ptr() as *const u32, public_input.len() as u32, &w, &b, 0, std::ptr::null_mut());
        let out = if rc == 0 {
            let mut bytes = vec![0u8; lmh_proof_postcard_size(p) as usize];
            lmh_proof_postcard(p, bytes.as_mut_ptr());
            proof_from_bytes(&bytes)
        } else {
            None
        };
        lmh_prover_free(p);
        out
    }
}

/// Decode the bytes written by `lmh_proof_postcard` into the reference's `Proof<F>`: same serde derive, same postcard.
pub fn proof_from_bytes(bytes: &[u8]) -> Option<backend::Proof<backend::F>> {
    postcard::from_bytes(bytes).ok()
}

#[cfg(test)]
mod tests {
    use super::*;

    /// External parity pin: `proof.bin` / `instance.bin` come from `python tools/write_proof.py` (a device proof, at the
    /// reference's default_whir_config parameters, of the witness of tests/golden/vectors_r01.json + the bytecode / public
    /// input it was proven for).
    #[test]
    fn reference_verifier_accepts_the_hip_proof() {
        let dir = std::env::var("LM_PROOF_DIR").unwrap_or_else(|_| "..".into());
        let proof_bytes = std::fs::read(format!("{dir}/proof.bin")).expect("proof.bin");
        let proof = proof_from_bytes(&proof_bytes).expect("postcard decode of Proof<F>");
        // instance.bin: u32 LE words [log_bytecode, ending_pc, n_public_input, bytecode_hash x 8, public_input.., bytecode rows x 16..]
        let inst = std::fs::read(format!("{dir}/instance.bin")).expect("instance.bin");
        let w: Vec<u32> = inst.chunks_exact(4).map(|c| u32::from_le_bytes(c.try_into().unwrap())).collect();
        let (log_bytecode, ending_pc, n_pub) = (w[0] as usize, w[1] as usize, w[2] as usize);
        let f = |x: u32| backend::F::new_monty(x);
        let hash: [backend::F; 8] = std::array::from_fn(|i| f(w[3 + i]));
        let public_input: Vec<backend::F> = w[11..11 + n_pub].iter().map(|&x| f(x)).collect();
        let rows: Vec<backend::F> = w[11 + n_pub..].iter().map(|&x| f(x)).collect();
        assert_eq!(rows.len(), 16 << log_bytecode);
        // verify_execution reads four things of `Bytecode` (crates/lean_vm/src/isa/bytecode.rs:18-30): instructions_multilinear,
        // hash, ending_pc and log_size() = log2_ceil(code.len()); `code` itself is only the VM's
        let bytecode = lean_vm::Bytecode {
            code: vec![Default::default(); 1 << log_bytecode],
            instructions_multilinear: rows,
            starting_frame_memory: 0,
            ending_pc,
            hash,
            function_locations: Default::default(),
            filepaths: Default::default(),
            source_code: Default::default(),
            pc_to_location: Vec::new(),
        };
        lean_prover::verify_execution::verify_execution(&bytecode, &public_input, proof).expect("reference verifier");
    }

    /// Second pin (round 4): the REAL-signature path.  `tests/golden/external_pin_xmss/{proof.bin, instance.zlib}` come from
    /// `python tools/write_proof.py tests/golden/external_pin_xmss --xmss`: the hand-assembled aggregation program verifying 40 XMSS
    /// signatures, run by the library's own leanVM (parallel batch on the device), proven at default_whir_config.  instance.zlib holds
    /// CANONICAL words, zlib-compressed: [log_bytecode, ending_pc, n_public_input, bytecode_hash x 8, public_input.., rows x 12 columns..].
    /// `make pin REFERENCE=<checkout>` runs both tests.
    #[test]
    fn reference_verifier_accepts_the_hip_proof_of_real_signatures() {
        use std::io::Read;
        let dir = std::env::var("LM_PROOF_DIR_XMSS").unwrap_or_else(|_| "../tests/golden/external_pin_xmss".into());
        let proof = proof_from_bytes(&std::fs::read(format!("{dir}/proof.bin")).expect("proof.bin")).expect("postcard decode of Proof<F>");
        let mut inst = Vec::new();
        flate2::read::ZlibDecoder::new(std::fs::File::open(format!("{dir}/instance.zlib")).expect("instance.zlib")).read_to_end(&mut inst).expect("zlib");
        let w: Vec<u32> = inst.chunks_exact(4).map(|c| u32::from_le_bytes(c.try_into().unwrap())).collect();
        let (log_bytecode, ending_pc, n_pub) = (w[0] as usize, w[1] as usize, w[2] as usize);
        let f = |x: u32| backend::F::new(x);  // canonical values
        let hash: [backend::F; 8] = std::array::from_fn(|i| f(w[3 + i]));
        let public_input: Vec<backend::F> = w[11..11 + n_pub].iter().map(|&x| f(x)).collect();
        assert_eq!(w.len() - 11 - n_pub, 12 << log_bytecode);
        // instructions_multilinear: 16 words per row, the 12 instruction columns followed by 4 zeros (lean_vm/src/isa/bytecode.rs)
        let mut rows = vec![backend::F::ZERO; 16 << log_bytecode];
        for (r, chunk) in w[11 + n_pub..].chunks_exact(12).enumerate() {
            for (c, &x) in chunk.iter().enumerate() {
                rows[16 * r + c] = f(x);
            }
        }
        let bytecode = lean_vm::Bytecode {
            code: vec![Default::default(); 1 << log_bytecode],
            instructions_multilinear: rows,
            starting_frame_memory: 0,
            ending_pc,
            hash,
            function_locations: Default::default(),
            filepaths: Default::default(),
            source_code: Default::default(),
            pc_to_location: Vec::new(),
        };
        lean_prover::verify_execution::verify_execution(&bytecode, &public_input, proof).expect("reference verifier");
    }
}
