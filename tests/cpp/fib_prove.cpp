// Driver of the compiled prover toyni_amd/csrc/host/fib_prover.hpp (reference: StarkProver::generate_proof, src/fibonacci.rs:99-310;
// its test module :401-456).  Run by tests/test_fib_prover_cpp.py (the proof it writes is checked by the CPU restatement of
// src/verifier.rs, tests/harness/fib_verifier.py) and by bench.py (extras.fib_prove_*: wall time per proof, warm).
//   fib_prove <trace_len> <seed> <reps> [proof.json] [--corrupt-row R] [--phases]
// stdout: one JSON line {"trace_len":..,"lde_size":..,"folds":..,"final_layer_size":..,"ms":[..],"phases":{..}}
#include "launch_dump.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../toyni_amd/csrc/host/fib_prover.hpp"

using namespace toyni::fib;

static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; ++i) { s[2 * i] = d[p[i] >> 4]; s[2 * i + 1] = d[p[i] & 15]; }
    return s;
}

static bool write_proof(const char* path, const Proof& pr) {
    FILE* f = std::fopen(path, "w");
    if (!f) return false;
    std::fprintf(f, "{\"trace_len\": %zu, \"lde_size\": %zu, \"trace_commitment\": \"%s\", \"quotient_commitment\": \"%s\",\n", pr.trace_len, pr.lde_size,
                 hex(pr.trace_commitment.data(), 32).c_str(), hex(pr.quotient_commitment.data(), 32).c_str());
    std::fprintf(f, " \"t_z\": %u, \"t_gz\": %u, \"t_ggz\": %u, \"q_z\": %u,\n \"fri_commitments\": [", pr.t_z, pr.t_gz, pr.t_ggz, pr.q_z);
    for (size_t i = 0; i < pr.fri_commitments.size(); ++i) std::fprintf(f, "%s\"%s\"", i ? ", " : "", hex(pr.fri_commitments[i].data(), 32).c_str());
    std::fprintf(f, "],\n \"fri_final_layer\": [");
    for (size_t i = 0; i < pr.fri_final_layer.size(); ++i) std::fprintf(f, "%s%u", i ? ", " : "", pr.fri_final_layer[i]);
    std::fprintf(f, "],\n \"query_indices\": [");
    for (size_t i = 0; i < pr.query_indices.size(); ++i) std::fprintf(f, "%s%u", i ? ", " : "", pr.query_indices[i]);
    std::fprintf(f, "],\n \"opening_groups\": [");
    for (size_t k = 0; k < pr.opening_groups.size(); ++k) {
        const auto& g = pr.opening_groups[k];
        std::fprintf(f, "%s[%zu, %s, [", k ? ", " : "", g.tree_leaves, g.salted ? "true" : "false");
        for (size_t i = 0; i < g.indices.size(); ++i) std::fprintf(f, "%s%u", i ? ", " : "", g.indices[i]);
        std::fprintf(f, "]]");
    }
    std::fprintf(f, "],\n \"opening_records\": \"%s\"}\n", hex(pr.opening_records.data(), pr.opening_records.size()).c_str());
    return std::fclose(f) == 0;
}

int main(int argc, char** argv) {
    struct DumpAtReturn { ~DumpAtReturn() { toyni_test_dump_launched_kernels(); } } dump_at_return;   // kernel-coverage guard of the test session
    if (argc < 4) { std::fprintf(stderr, "usage: %s <trace_len> <seed> <reps> [proof.json] [--corrupt-row R] [--phases]\n", argv[0]); return 2; }
    const size_t n = std::strtoull(argv[1], nullptr, 0);
    const uint64_t seed = std::strtoull(argv[2], nullptr, 0);
    const int reps = std::atoi(argv[3]);
    const char* out_path = nullptr;
    long corrupt = -1;
    bool phases = false;
    for (int i = 4; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--corrupt-row") && i + 1 < argc) corrupt = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--phases")) phases = true;
        else out_path = argv[i];
    }
    int count = 0;
    if (toyni_device_count(&count) != 0 || count <= 0) { std::printf("{\"gpu\": false}\n"); return 0; }   // self-skip like src/ntt.rs:265-268
    auto trace = fibonacci_trace(n);
    if (corrupt >= 0 && (size_t)corrupt < n) trace[(size_t)corrupt] = addmod(trace[(size_t)corrupt], 1);     // src/fibonacci.rs:430-442
    uint8_t key[32];
    for (int i = 0; i < 32; ++i) key[i] = (uint8_t)((seed * 0x9E3779B97F4A7C15ull + 0x1234567u * (unsigned)i) >> (8 * (i % 8)));   // a TEST key (a prover draws it from the OS)
    Prover prover(n);
    Proof proof;
    std::string err = prover.generate_proof(trace.data(), key, proof);   // warm: contexts, tables, buffers
    if (!err.empty()) { std::printf("{\"gpu\": true, \"error\": \"%s\"}\n", err.c_str()); return 1; }
    std::string ms = "[";
    for (int r = 0; r < reps; ++r) {
        key[0] = (uint8_t)(key[0] + 1);
        const auto t0 = std::chrono::steady_clock::now();
        err = prover.generate_proof(trace.data(), key, proof);
        const auto t1 = std::chrono::steady_clock::now();
        if (!err.empty()) { std::printf("{\"gpu\": true, \"error\": \"%s\"}\n", err.c_str()); return 1; }
        char buf[64];
        std::snprintf(buf, sizeof buf, "%s%.4f", r ? ", " : "", std::chrono::duration<double, std::milli>(t1 - t0).count());
        ms += buf;
    }
    ms += "]";
    PhaseTimes pt;
    if (phases) {
        key[1] = (uint8_t)(key[1] + 1);
        err = prover.generate_proof(trace.data(), key, proof, &pt);
        if (!err.empty()) { std::printf("{\"gpu\": true, \"error\": \"%s\"}\n", err.c_str()); return 1; }
    }
    if (out_path && !write_proof(out_path, proof)) { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; }
    std::printf("{\"gpu\": true, \"trace_len\": %zu, \"lde_size\": %zu, \"folds\": %zu, \"final_layer_size\": %zu, \"ms\": %s, "
                "\"phases\": {\"1_interpolate_mask_lde_commit\": %.4f, \"2_constraint_quotient_commit\": %.4f, \"3_transcript_ood\": %.4f, \"5_deep\": %.4f, "
                "\"6_fri_fold_commit\": %.4f, \"7_queries\": %.4f}, \"proof_bytes\": %zu, "
                "\"pcie\": {\"h2d_bytes\": %zu, \"d2h_bytes\": %zu, \"h2d_copies\": %zu, \"d2h_copies\": %zu, \"pinned_root_writes\": %zu}}\n",
                proof.trace_len, proof.lde_size, prover.folds(), prover.final_layer_size(), ms.c_str(), pt.interpolate_lde_commit, pt.quotient_commit,
                pt.transcript_ood, pt.deep, pt.fri, pt.queries,
                proof.opening_records.size() + 32 * (2 + proof.fri_commitments.size()) + 4 * (4 + proof.fri_final_layer.size()),
                prover.traffic().h2d_bytes, prover.traffic().d2h_bytes, prover.traffic().h2d_copies, prover.traffic().d2h_copies, prover.traffic().pinned_root_writes);
    return 0;
}
