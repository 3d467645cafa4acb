// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).
//
// prove_execution / verify_execution AFTER witness generation (crates/lean_prover/src/prove_execution.rs:47-274,
// verify_execution.rs:14-233), i.e. the slice that starts from the execution trace (memory, access counters, tables)
// and ends with the proof.  The VM interpreter / trace builder that precede it are out of scope.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include "logup_oracle.hpp"

namespace orc {

static const uint32_t SNARK_DOMAIN_SEP_CANON[8] = {130704175, 1303721200, 493664240, 1035493700,
                                                   2063844858, 1410214009, 1938905908, 1696767928};  // lean_prover/src/lib.rs:30-32
static const size_t N_INSTRUCTION_COLUMNS = 12, N_RUNTIME_COLUMNS = 8, COL_PC = 0;

struct ExecutionInput {
    size_t log_inv_rate;
    std::vector<uint32_t> public_input;
    uint32_t bytecode_hash[8];
    const uint32_t* bytecode;  // 2^log_bytecode x 16
    size_t log_bytecode, ending_pc;
    const uint32_t *memory, *memory_acc;
    size_t log_memory;
    const uint32_t* bytecode_acc;
    size_t public_memory_size;
    VmTableTrace tables[3];  // indexed by table id (execution, extension_op, poseidon16)
};

static inline std::vector<int> sort_tables_by_height(const size_t log_rows[3]) {  // tables/table_trait.rs:66-70 (stable)
    std::vector<int> o{0, 1, 2};
    std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return log_rows[a] > log_rows[b]; });
    return o;
}
static inline size_t log2_ceil(size_t x) {
    size_t l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}
// stacked_pcs.rs:183-196
static inline size_t compute_stacked_n_vars(size_t log_mem, size_t log_bc, const size_t log_rows[3]) {
    size_t mx = std::max(log_rows[0], std::max(log_rows[1], log_rows[2]));
    size_t total = ((size_t)2 << log_mem) + ((size_t)1 << std::max(log_bc, mx));
    for (int t = 0; t < 3; t++) total += air_n_columns(t) << log_rows[t];
    return log2_ceil(total);
}
static inline std::vector<EF> from_end(const std::vector<EF>& v, size_t n) { return std::vector<EF>(v.end() - n, v.end()); }
// poly/src/mle/mle_custom.rs:4-19
static inline EF mle_of_zeros_then_ones(size_t n_zeros, const EF* point, size_t n) {
    size_t n_values = (size_t)1 << n;
    if (n_zeros == 0) return ef_one();
    if (n_zeros == n_values) return ef_zero();
    size_t half = n_values / 2;
    if (n_zeros < half) return ef_add(ef_mul(ef_sub(ef_one(), point[0]), mle_of_zeros_then_ones(n_zeros, point + 1, n - 1)), point[0]);
    return ef_mul(point[0], mle_of_zeros_then_ones(n_zeros - half, point + 1, n - 1));
}
// utils/src/multilinear.rs:67-74
static inline EF mle_of_01234567_etc(const EF* point, size_t n) {
    if (n == 0) return ef_zero();
    EF e = mle_of_01234567_etc(point + 1, n - 1);
    EF hi = ef_add(e, ef_from_base(to_monty((uint32_t)((size_t)1 << (n - 1)))));
    return ef_add(ef_mul(ef_sub(ef_one(), point[0]), e), ef_mul(point[0], hi));
}
static inline EF finger_print_ef(uint32_t domsep, const std::vector<EF>& data, const EF* alphas) {
    EF s = ef_mul_base(alphas[15], to_monty(domsep));
    for (size_t j = 0; j < data.size(); j++) s = ef_add(s, ef_mul(alphas[j], data[j]));
    return s;
}

struct TableStatement {  // one entry of CommittedStatements (tables/table_trait.rs:11-13)
    std::vector<EF> point;
    std::map<size_t, EF> eq_values, next_values;
};

// stacked_pcs_global_statements (stacked_pcs.rs:40-97)
static inline std::vector<SparseStatement> stacked_pcs_global_statements(size_t stacked_n_vars, size_t log_mem, size_t log_bc, size_t ending_pc,
                                                                         std::vector<SparseStatement> previous, const size_t log_rows[3],
                                                                         const std::vector<TableStatement> committed[3]) {
    std::vector<SparseStatement> g = std::move(previous);
    size_t mx = std::max(log_rows[0], std::max(log_rows[1], log_rows[2]));
    size_t offset = ((size_t)2 << log_mem) + ((size_t)1 << std::max(log_bc, mx));
    for (int t : sort_tables_by_height(log_rows)) {
        size_t nv = log_rows[t];
        auto unique_value = [&](size_t index, EF value) {
            SparseStatement s;
            s.total_num_variables = stacked_n_vars;
            s.values.push_back({index, value});
            return s;
        };
        if (t == AIR_EXECUTION) {
            g.push_back(unique_value(offset + (COL_PC << nv), ef_const(0)));  // STARTING_PC = 0
            g.push_back(unique_value(offset + ((COL_PC + 1) << nv) - 1, ef_const((uint32_t)ending_pc)));
        }
        for (const TableStatement& st : committed[t]) {
            if (!st.next_values.empty()) {
                SparseStatement s;
                s.total_num_variables = stacked_n_vars;
                s.point = st.point;
                s.is_next = true;
                for (auto& kv : st.next_values) s.values.push_back({(offset >> nv) + kv.first, kv.second});
                g.push_back(s);
            }
            SparseStatement s;
            s.total_num_variables = stacked_n_vars;
            s.point = st.point;
            for (auto& kv : st.eq_values) s.values.push_back({(offset >> nv) + kv.first, kv.second});
            g.push_back(s);
        }
        offset += air_n_columns(t) << nv;
    }
    return g;
}

static inline WhirConfigBuilder default_whir_config(size_t log_inv_rate) {  // lean_prover/src/lib.rs:34-50 (proven regime)
    WhirConfigBuilder b;
    b.starting_log_inv_rate = log_inv_rate;
    return b;
}

static inline void fs_preamble_prover(ProverState& ps, const ExecutionInput& in) {  // prove_execution.rs:47-63
    ps.observe_scalars(in.public_input.data(), in.public_input.size());
    uint32_t dom[8], h[8];
    for (int i = 0; i < 8; i++) dom[i] = to_monty(SNARK_DOMAIN_SEP_CANON[i]);
    compress_pair(in.bytecode_hash, dom, h);
    ps.observe_scalars(h, 8);
    std::vector<uint32_t> dims{to_monty((uint32_t)in.log_inv_rate), to_monty((uint32_t)in.log_memory), to_monty((uint32_t)in.public_input.size())};
    for (int t = 0; t < 3; t++) dims.push_back(to_monty((uint32_t)in.tables[t].log_rows));
    ps.add_base_scalars(dims.data(), dims.size());
}

// builder: default_whir_config(rate) unless the test overrides PoW/security to keep the oracle fast
// ORC_STAGE_TIMES=1: wall clock per stage on stderr (where the CPU baseline spends its time)
struct OrcStageClock {
    bool on = getenv("ORC_STAGE_TIMES") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* name) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "# oracle stage %-24s %9.1f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
static inline void prove_execution(ProverState& ps, const ExecutionInput& in, const WhirConfigBuilder& builder) {
    OrcStageClock clk;
    fs_preamble_prover(ps, in);
    size_t log_rows[3] = {in.tables[0].log_rows, in.tables[1].log_rows, in.tables[2].log_rows};
    std::vector<int> order = sort_tables_by_height(log_rows);
    // stack_polynomials_and_commit (stacked_pcs.rs:99-157)
    size_t stacked_n_vars = compute_stacked_n_vars(in.log_memory, in.log_bytecode, log_rows);
    std::vector<uint32_t> poly((size_t)1 << stacked_n_vars, 0);
    size_t mem = (size_t)1 << in.log_memory;
    std::memcpy(poly.data(), in.memory, mem * 4);
    std::memcpy(poly.data() + mem, in.memory_acc, mem * 4);
    size_t off = 2 * mem;
    std::memcpy(poly.data() + off, in.bytecode_acc, ((size_t)4) << in.log_bytecode);
    off += std::max((size_t)1 << log_rows[order[0]], (size_t)1 << in.log_bytecode);
    for (int t : order)
        for (size_t c = 0; c < air_n_columns(t); c++) {
            std::memcpy(poly.data() + off, in.tables[t].col(c), ((size_t)4) << log_rows[t]);
            off += (size_t)1 << log_rows[t];
        }
    WhirConfig cfg = WhirConfig::make(builder, stacked_n_vars);
    clk.mark("stack");
    Witness wit = whir_commit(cfg, ps, poly.data(), off);
    clk.mark("whir_commit");
    // logup (prove_execution.rs:123-150, logup.rs)
    EF logup_c = ps.sample();
    ps.duplex();
    std::vector<EF> logup_alphas = ps.sample_vec(4);
    std::vector<EF> aeq = eq_table(logup_alphas.data(), 4, ef_one());
    std::vector<VmTableTrace> sorted;
    for (int t : order) sorted.push_back(in.tables[t]);
    std::vector<uint32_t> nums;
    std::vector<EF> dens;
    logup_fill(in.memory, in.memory_acc, in.log_memory, in.bytecode, in.bytecode_acc, in.log_bytecode, sorted, logup_c, aeq.data(), nums, dens);
    size_t gkr_n_vars = log2_ceil(nums.size());
    EF quotient, cn, cd;
    std::vector<EF> gkr_point;
    clk.mark("logup_fill");
    gkr_prove(ps, nums.data(), dens.data(), gkr_n_vars, quotient, gkr_point, cn, cd);
    clk.mark("logup_gkr");
    if (!ef_eq(quotient, ef_zero())) throw std::runtime_error("logup sum != 0 (inconsistent witness)");
    std::vector<EF> mem_pt = from_end(gkr_point, in.log_memory);
    EF value_memory_acc = mle_eval_base(in.memory_acc, in.log_memory, mem_pt.data());
    ps.add_extension_scalars({value_memory_acc});
    EF value_memory = mle_eval_base(in.memory, in.log_memory, mem_pt.data());
    ps.add_extension_scalars({value_memory});
    std::vector<EF> bc_pt = from_end(gkr_point, in.log_bytecode);
    EF value_bytecode_acc = mle_eval_base(in.bytecode_acc, in.log_bytecode, bc_pt.data());
    ps.add_extension_scalars({value_bytecode_acc});
    std::map<size_t, EF> columns_values[3];
    EF bus_num[3], bus_den[3];
    for (int t : order) {
        const VmTableTrace& tr = in.tables[t];
        std::vector<EF> ip = from_end(gkr_point, tr.log_rows);
        auto ev = [&](size_t col) { return mle_eval_base(tr.col(col), tr.log_rows, ip.data()); };
        const VmTableDef def = vm_table_def(t);
        if (t == AIR_EXECUTION) {
            EF epc = ev(COL_PC);
            ps.add_extension_scalars({epc});
            columns_values[t][COL_PC] = epc;
            std::vector<EF> ie;
            for (size_t k = 0; k < N_INSTRUCTION_COLUMNS; k++) ie.push_back(ev(N_RUNTIME_COLUMNS + k));
            ps.add_extension_scalars(ie);
            for (size_t k = 0; k < N_INSTRUCTION_COLUMNS; k++) columns_values[t][N_RUNTIME_COLUMNS + k] = ie[k];
        }
        EF esel = ev(def.bus.selector);
        if (def.bus.pull) esel = ef_neg(esel);  // * direction.to_field_flag()
        ps.add_extension_scalars({esel});
        std::vector<EF> bd;
        for (size_t c : def.bus.data) bd.push_back(ev(c));
        EF edata = ef_add(logup_c, finger_print_ef(1, bd, aeq.data()));
        ps.add_extension_scalars({edata});
        bus_num[t] = esel;
        bus_den[t] = edata;
        for (const LookupIntoMemory& lk : def.lookups) {
            EF ie = ev(lk.index);
            ps.add_extension_scalars({ie});
            columns_values[t][lk.index] = ie;
            for (size_t c : lk.values) {
                EF ve = ev(c);
                ps.add_extension_scalars({ve});
                columns_values[t][c] = ve;
            }
        }
    }
    std::vector<TableStatement> committed[3];
    for (int t = 0; t < 3; t++) committed[t].push_back({from_end(gkr_point, log_rows[t]), columns_values[t], {}});
    clk.mark("column_evaluations");
    // AIR (prove_execution.rs:152-223)
    EF bus_beta = ps.sample();
    ps.duplex();
    EF air_alpha = ps.sample();
    ps.duplex();
    EF air_eta = ps.sample();
    AirExtra ex;
    ex.bus_beta = bus_beta;
    ex.logup_alphas_eq_poly = aeq;
    ex.alpha_powers.resize(101);
    ex.alpha_powers[0] = ef_one();
    for (int i = 1; i < 101; i++) ex.alpha_powers[i] = ef_mul(ex.alpha_powers[i - 1], air_alpha);
    std::vector<AirSession> sessions;
    for (int t : order) {
        const VmTableTrace& tr = in.tables[t];
        AirSession s;
        s.table = t;
        s.n_vars = tr.log_rows;
        s.eq_factor = from_end(gkr_point, tr.log_rows);
        EF dir = vm_table_def(t).bus.pull ? ef_neg(ef_one()) : ef_one();
        s.sum = ef_add(ef_mul(bus_num[t], dir), ef_mul(bus_beta, ef_sub(bus_den[t], logup_c)));
        s.mmf = ef_one();
        s.extra = ex;
        size_t n = (size_t)1 << tr.log_rows, nc = air_n_columns(t), ns = air_n_shift(t);
        s.cols.resize(nc + ns);
        for (size_t c = 0; c < nc; c++) {
            s.cols[c].resize(n);
            for (size_t i = 0; i < n; i++) s.cols[c][i] = ef_from_base(tr.col(c)[i]);
        }
        for (size_t c = 0; c < ns; c++) {
            std::vector<uint32_t> sh = shifted_column(tr.col(c), n);
            s.cols[nc + c].resize(n);
            for (size_t i = 0; i < n; i++) s.cols[nc + c][i] = ef_from_base(sh[i]);
        }
        sessions.push_back(std::move(s));
    }
    std::vector<EF> air_point = prove_batched_air_sumcheck(ps, sessions, air_eta);
    for (size_t k = 0; k < order.size(); k++) {
        int t = order[k];
        std::vector<EF> ce = sessions[k].final_column_evals();
        ps.add_extension_scalars(ce);
        TableStatement st;
        for (size_t j = 0; j < log_rows[t]; j++) st.point.push_back(air_point[air_point.size() - 1 - j]);  // natural_ordering_point_for_session
        for (size_t c = 0; c < air_n_columns(t); c++) st.eq_values[c] = ce[c];
        for (size_t c = 0; c < air_n_shift(t); c++) st.next_values[c] = ce[air_n_columns(t) + c];
        committed[t].push_back(st);
    }
    // public memory + global statements (:225-260)
    size_t lpm = log2_ceil(in.public_memory_size);
    std::vector<EF> pm_pt = ps.sample_vec(lpm);
    EF pm_eval = mle_eval_base(in.memory, lpm, pm_pt.data());
    auto mk = [&](std::vector<EF> point, std::vector<SparseValue> vals) {
        SparseStatement s;
        s.total_num_variables = stacked_n_vars;
        s.point = std::move(point);
        s.values = std::move(vals);
        return s;
    };
    std::vector<SparseStatement> prev;
    prev.push_back(mk(mem_pt, {{0, value_memory}, {1, value_memory_acc}}));
    prev.push_back(mk(pm_pt, {{0, pm_eval}}));
    prev.push_back(mk(bc_pt, {{(2 * mem) >> in.log_bytecode, value_bytecode_acc}}));
    std::vector<SparseStatement> global = stacked_pcs_global_statements(stacked_n_vars, in.log_memory, in.log_bytecode, in.ending_pc, prev, log_rows, committed);
    clk.mark("air_sumcheck+statements");
    whir_prove(cfg, ps, global, std::move(wit), poly.data());
    clk.mark("whir_open");
}

// verify_execution (verify_execution.rs:14-233) + verify_generic_logup (logup.rs:326-493).  Throws on failure.
static inline void verify_execution(VerifierState& vs, const std::vector<uint32_t>& public_input, const uint32_t bytecode_hash[8],
                                    const uint32_t* bytecode, size_t log_bytecode, size_t ending_pc, const WhirConfigBuilder* builder_override) {
    vs.observe_scalars(public_input.data(), public_input.size());
    uint32_t dom[8], h[8];
    for (int i = 0; i < 8; i++) dom[i] = to_monty(SNARK_DOMAIN_SEP_CANON[i]);
    compress_pair(bytecode_hash, dom, h);
    vs.observe_scalars(h, 8);
    std::vector<uint32_t> dims = vs.next_base_scalars_vec(6);
    size_t log_inv_rate = from_monty(dims[0]), log_memory = from_monty(dims[1]);
    if (from_monty(dims[2]) != public_input.size()) throw std::runtime_error("InvalidProof (public input length)");
    size_t log_rows[3] = {from_monty(dims[3]), from_monty(dims[4]), from_monty(dims[5])};
    for (int t = 0; t < 3; t++)
        if (log_rows[t] < 8 || log_rows[t] > 24) throw std::runtime_error("InvalidProof (table size)");
    size_t mx = std::max(log_rows[0], std::max(log_rows[1], log_rows[2]));
    if (log_memory < std::max(mx, log_bytecode) || log_memory < 16 || log_memory > 26 || log_bytecode < 8)
        throw std::runtime_error("InvalidProof (memory size)");
    if (log_memory < log_rows[0] || log_rows[0] < mx) throw std::runtime_error("InvalidProof (execution table must be the largest)");
    std::vector<uint32_t> public_memory = public_input;
    size_t pms = 1;
    while (pms < public_memory.size()) pms <<= 1;
    public_memory.resize(pms, 0);
    WhirConfigBuilder builder = builder_override ? *builder_override : default_whir_config(log_inv_rate);
    size_t stacked_n_vars = compute_stacked_n_vars(log_memory, log_bytecode, log_rows);
    WhirConfig cfg = WhirConfig::make(builder, stacked_n_vars);
    ParsedCommitment pc = parse_commitment(vs, stacked_n_vars, cfg.commitment_ood_samples);
    EF logup_c = vs.sample();
    vs.duplex();
    std::vector<EF> alphas = vs.sample_vec(4);
    std::vector<EF> aeq = eq_table(alphas.data(), 4, ef_one());
    // ---- verify_generic_logup ----
    std::vector<int> order = sort_tables_by_height(log_rows);
    size_t total_active = ((size_t)1 << log_memory) + std::max((size_t)1 << log_bytecode, (size_t)1 << log_rows[order[0]]) + ((size_t)1 << log_rows[0]);
    for (int t = 0; t < 3; t++) {
        size_t ncols = 1;
        for (auto& lk : vm_table_def(t).lookups) ncols += lk.values.size();
        total_active += ncols << log_rows[t];
    }
    size_t gkr_n_vars = log2_ceil(total_active);
    EF quotient, num_value, den_value;
    std::vector<EF> gp;
    gkr_verify(vs, gkr_n_vars, quotient, gp, num_value, den_value);
    if (!ef_eq(quotient, ef_zero())) throw std::runtime_error("InvalidProof (logup sum)");
    EF rn = ef_zero(), rd = ef_zero();
    auto pref_at = [&](size_t offset, size_t log_h) {
        size_t n_missing = gkr_n_vars - log_h;
        EF acc = ef_one();
        for (size_t j = 0; j < n_missing; j++) {
            bool bit = ((offset >> log_h) >> (n_missing - 1 - j)) & 1;
            acc = ef_mul(acc, bit ? gp[j] : ef_sub(ef_one(), gp[j]));
        }
        return acc;
    };
    std::vector<EF> mem_pt = from_end(gp, log_memory);
    EF pref = pref_at(0, log_memory);
    EF value_memory_acc = vs.next_extension_scalars_vec(1)[0];
    rn = ef_sub(rn, ef_mul(pref, value_memory_acc));
    EF value_memory = vs.next_extension_scalars_vec(1)[0];
    EF value_index = mle_of_01234567_etc(mem_pt.data(), mem_pt.size());
    rd = ef_add(rd, ef_mul(pref, ef_sub(logup_c, finger_print_ef(0, {value_memory, value_index}, aeq.data()))));
    size_t offset = (size_t)1 << log_memory;
    size_t log_bc_padded = std::max(log_bytecode, log_rows[order[0]]);
    std::vector<EF> bc_pt = from_end(gp, log_bytecode);
    pref = pref_at(offset, log_bytecode);
    EF pref_padded = pref_at(offset, log_bc_padded);
    EF value_bytecode_acc = vs.next_extension_scalars_vec(1)[0];
    rn = ef_sub(rn, ef_mul(pref, value_bytecode_acc));
    EF bc_index_value = mle_of_01234567_etc(bc_pt.data(), bc_pt.size());
    std::vector<EF> bcp = bc_pt;
    for (int j = 0; j < 4; j++) bcp.push_back(alphas[j]);  // from_end(alphas, log2_ceil(12)) with 4 alphas
    EF bc_value = mle_eval_base(bytecode, log_bytecode + 4, bcp.data());
    // alphas[..len - 4] is empty here (4 alphas): product over an empty range = 1
    rd = ef_add(rd, ef_mul(pref, ef_sub(logup_c, ef_add(ef_add(bc_value, ef_mul(bc_index_value, aeq[N_INSTRUCTION_COLUMNS])),
                                                         ef_mul_base(aeq[15], to_monty(2))))));
    std::vector<EF> padpt = from_end(gp, log_bc_padded);
    rd = ef_add(rd, ef_mul(pref_padded, mle_of_zeros_then_ones((size_t)1 << log_bytecode, padpt.data(), padpt.size())));
    offset += (size_t)1 << log_bc_padded;
    std::map<size_t, EF> columns_values[3];
    EF bus_num[3], bus_den[3];
    for (int t : order) {
        size_t lr = log_rows[t];
        const VmTableDef def = vm_table_def(t);
        if (t == AIR_EXECUTION) {
            EF epc = vs.next_extension_scalars_vec(1)[0];
            columns_values[t][COL_PC] = epc;
            std::vector<EF> ie = vs.next_extension_scalars_vec(N_INSTRUCTION_COLUMNS);
            for (size_t k = 0; k < N_INSTRUCTION_COLUMNS; k++) columns_values[t][N_RUNTIME_COLUMNS + k] = ie[k];
            EF p = pref_at(offset, lr);
            rn = ef_add(rn, p);
            std::vector<EF> d = ie;
            d.push_back(epc);
            rd = ef_add(rd, ef_mul(p, ef_sub(logup_c, finger_print_ef(2, d, aeq.data()))));
            offset += (size_t)1 << lr;
        }
        EF esel = vs.next_extension_scalars_vec(1)[0];
        EF p = pref_at(offset, lr);
        rn = ef_add(rn, ef_mul(p, esel));
        EF edata = vs.next_extension_scalars_vec(1)[0];
        rd = ef_add(rd, ef_mul(p, edata));
        bus_num[t] = esel;
        bus_den[t] = edata;
        offset += (size_t)1 << lr;
        for (const LookupIntoMemory& lk : def.lookups) {
            EF ie = vs.next_extension_scalars_vec(1)[0];
            columns_values[t][lk.index] = ie;
            for (size_t i = 0; i < lk.values.size(); i++) {
                EF ve = vs.next_extension_scalars_vec(1)[0];
                columns_values[t][lk.values[i]] = ve;
                EF pp = pref_at(offset, lr);
                rn = ef_add(rn, pp);
                rd = ef_add(rd, ef_mul(pp, ef_sub(logup_c, finger_print_ef(0, {ve, ef_add(ie, ef_const((uint32_t)i))}, aeq.data()))));
                offset += (size_t)1 << lr;
            }
        }
    }
    rd = ef_add(rd, mle_of_zeros_then_ones(offset, gp.data(), gp.size()));
    if (!ef_eq(rn, num_value)) throw std::runtime_error("InvalidProof (logup numerators)");
    if (!ef_eq(rd, den_value)) throw std::runtime_error("InvalidProof (logup denominators)");
    std::vector<TableStatement> committed[3];
    for (int t = 0; t < 3; t++) committed[t].push_back({from_end(gp, log_rows[t]), columns_values[t], {}});
    // ---- AIR ----
    EF bus_beta = vs.sample();
    vs.duplex();
    EF air_alpha = vs.sample();
    vs.duplex();
    EF eta = vs.sample();
    AirExtra ex;
    ex.bus_beta = bus_beta;
    ex.logup_alphas_eq_poly = aeq;
    ex.alpha_powers.resize(101);
    ex.alpha_powers[0] = ef_one();
    for (int i = 1; i < 101; i++) ex.alpha_powers[i] = ef_mul(ex.alpha_powers[i - 1], air_alpha);
    EF initial_sum = ef_zero(), ep = ef_one();
    std::vector<EF> eta_p;
    for (int t : order) {
        EF dir = vm_table_def(t).bus.pull ? ef_neg(ef_one()) : ef_one();
        EF bfv = ef_add(ef_mul(bus_num[t], dir), ef_mul(bus_beta, ef_sub(bus_den[t], logup_c)));
        initial_sum = ef_add(initial_sum, ef_mul(ep, bfv));
        eta_p.push_back(ep);
        ep = ef_mul(ep, eta);
    }
    size_t n_max = log_rows[order[0]], max_full_degree = 11;
    EF target = initial_sum;
    std::vector<EF> ap;
    for (size_t r = 0; r < n_max; r++) {
        std::vector<EF> coeffs = vs.next_sumcheck_polynomial(max_full_degree + 1, target, nullptr);
        EF c = vs.sample();
        ap.push_back(c);
        target = poly_eval(coeffs, c);
    }
    EF mine = ef_zero();
    for (size_t k = 0; k < order.size(); k++) {
        int t = order[k];
        size_t nct = air_n_columns(t) + air_n_shift(t);
        std::vector<EF> ce = vs.next_extension_scalars_vec(nct);
        EF cev = air_eval(t, ce.data(), ex);
        std::vector<EF> bus_point = from_end(gp, log_rows[t]);
        std::vector<EF> nat;
        for (size_t j = 0; j < log_rows[t]; j++) nat.push_back(ap[ap.size() - 1 - j]);
        EF eqv = ef_one();
        for (size_t j = 0; j < log_rows[t]; j++)
            eqv = ef_mul(eqv, ef_add(ef_mul(bus_point[j], nat[j]), ef_mul(ef_sub(ef_one(), bus_point[j]), ef_sub(ef_one(), nat[j]))));
        EF kt = ef_one();
        for (size_t j = 0; j < n_max - log_rows[t]; j++) kt = ef_mul(kt, ap[j]);
        mine = ef_add(mine, ef_mul(ef_mul(ef_mul(eta_p[k], kt), eqv), cev));
        TableStatement st;
        st.point = nat;
        for (size_t c = 0; c < air_n_columns(t); c++) st.eq_values[c] = ce[c];
        for (size_t c = 0; c < air_n_shift(t); c++) st.next_values[c] = ce[air_n_columns(t) + c];
        committed[t].push_back(st);
    }
    if (!ef_eq(mine, target)) throw std::runtime_error("InvalidProof (air final value)");
    size_t lpm = log2_ceil(public_memory.size());
    std::vector<EF> pm_pt = vs.sample_vec(lpm);
    EF pm_eval = mle_eval_base(public_memory.data(), lpm, pm_pt.data());
    auto mk = [&](std::vector<EF> point, std::vector<SparseValue> vals) {
        SparseStatement s;
        s.total_num_variables = stacked_n_vars;
        s.point = std::move(point);
        s.values = std::move(vals);
        return s;
    };
    std::vector<SparseStatement> prev;
    prev.push_back(mk(mem_pt, {{0, value_memory}, {1, value_memory_acc}}));
    prev.push_back(mk(pm_pt, {{0, pm_eval}}));
    prev.push_back(mk(bc_pt, {{((size_t)2 << log_memory) >> log_bytecode, value_bytecode_acc}}));
    std::vector<SparseStatement> global = stacked_pcs_global_statements(stacked_n_vars, log_memory, log_bytecode, ending_pc, prev, log_rows, committed);
    whir_verify(cfg, vs, pc, global);
    if (vs.off != vs.transcript.size()) throw std::runtime_error("trailing transcript data");
    if (vs.merkle_idx != vs.merkle_openings.size()) throw std::runtime_error("unused merkle openings");
}

}  // namespace orc
