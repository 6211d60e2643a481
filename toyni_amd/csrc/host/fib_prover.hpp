// C++ counterpart of `StarkProver::generate_proof(use_gpu)` (reference src/fibonacci.rs:99-310) over the C ABI of
// include/toyni_hip.h -- BASELINE configs[2] (Fibonacci AIR, blowup 32) with a COMPILED caller (SURVEY.md section 2 asks for
// "a minimal C++ harness counterpart"; round 2's caller was a Python harness whose 4.4 ms included interpreter time).
//
// Same protocol, step for step (masking, trace / quotient / DEEP layers, salted Merkle commitments, the Fiat-Shamir transcript of
// src/transcript.rs, the fold loop with its per-round root, 44 queries); the heavy steps are the library's device calls, chosen as
// SURVEY.md F3 / F5 prescribe (the reference's O(n^3) interpolation and O(N d) Horner LDE are infeasible at 2^16 rows):
//   interpolate              toyni_ntt_device (inverse, size n)                 src/fibonacci.rs:110-111
//   LDE on the coset         toyni_lde_device (padding implied)                 :124-128
//   commitments              toyni_merkle_commit_device                         :129-130, :153-154, :205-208
//   constraint / quotient    toyni_fib_quotient_device + the two coset INTTs    :133-151
//   OOD evaluations          toyni_poly_eval_device                             :165-168
//   DEEP layer               toyni_fib_deep_device                              :186-198
//   fold loop                toyni_fri_commit_phase_device (transcript = callback) :220-245
//   openings                 toyni_merkle_open_device                           :249-295
// Everything between those calls stays on the device.  What crosses PCIe per proof (measured: profiles/r03_fib_prove_memcopy.txt):
//   up:   the trace column (4 n bytes), 140 + 140 masked coefficients, the query positions (~1 700 x 4 bytes)
//   down: 140 coefficients (to mask them), 3 + rounds roots of 32 bytes, 4 OOD values, the final layer, the opening records
// Salts and mask coefficients come from a ChaCha20 keystream keyed by the caller (the reference: rand::thread_rng, :117,:341).
// The proof is checked by tests/harness/fib_verifier.py (a CPU restatement of src/verifier.rs) on its serialized form.
// No CPU fallback: without the library's device path every call returns its status.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../prover_kernels.hpp"   // chacha20_block, sha256_compress (plain C++ bodies shared with the device code)
#include "toyni_hip.h"

namespace toyni {
namespace fib {

constexpr uint32_t P = 2013265921u;
constexpr unsigned NUM_QUERIES = 44;                    // src/fibonacci.rs:11
constexpr unsigned LOG_BLOWUP = 5;                      // BLOWUP = 32, :14
constexpr uint32_t COSET_SHIFT = 7;                     // :16
constexpr unsigned MASK_DEGREE = 3 * NUM_QUERIES + 8;   // :19

inline uint32_t mulmod(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); }
inline uint32_t addmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % P); }
inline uint32_t submod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + P - b) % P); }
inline uint32_t powmod(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; }
    return r;
}
inline uint32_t root_of_unity(unsigned log_n) { return powmod(440564289u, (uint64_t)1 << (27 - log_n)); }   // src/babybear.rs:118-126

// SHA-256 of a byte string on the host (FIPS 180-4; the compression function is the one the device kernels use)
inline std::array<uint8_t, 32> sha256(const uint8_t* data, size_t len) {
    Sha256State st = sha256_init();
    uint32_t w[16];
    size_t off = 0;
    auto load = [&](const uint8_t* b) { for (int j = 0; j < 16; ++j) w[j] = (uint32_t)b[4 * j] << 24 | (uint32_t)b[4 * j + 1] << 16 | (uint32_t)b[4 * j + 2] << 8 | b[4 * j + 3]; };
    for (; off + 64 <= len; off += 64) { load(data + off); sha256_compress(st, w); }
    uint8_t tail[128] = {0};
    const size_t rem = len - off;
    std::memcpy(tail, data + off, rem);
    tail[rem] = 0x80;
    const size_t blocks = rem + 9 <= 64 ? 1 : 2;
    const uint64_t bits = (uint64_t)len * 8;
    for (int j = 0; j < 8; ++j) tail[blocks * 64 - 1 - j] = (uint8_t)(bits >> (8 * j));
    for (size_t b = 0; b < blocks; ++b) { load(tail + 64 * b); sha256_compress(st, w); }
    std::array<uint8_t, 32> out;
    for (int j = 0; j < 8; ++j) { out[4 * j] = (uint8_t)(st.h[j] >> 24); out[4 * j + 1] = (uint8_t)(st.h[j] >> 16); out[4 * j + 2] = (uint8_t)(st.h[j] >> 8); out[4 * j + 3] = (uint8_t)st.h[j]; }
    return out;
}

// src/transcript.rs:12-72
class Transcript {
   public:
    Transcript() { const char* tag = "toyni-stark-v1"; state_.assign(tag, tag + 14); }
    void absorb(const uint8_t* d, size_t n) { state_.insert(state_.end(), d, d + n); }
    void absorb_field(uint32_t v) { uint8_t b[8] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24), 0, 0, 0, 0}; absorb(b, 8); }
    uint64_t squeeze_u64() {
        const auto h = sha256(state_.data(), state_.size());
        state_.assign(h.begin(), h.end());
        uint64_t v = 0;
        for (int j = 7; j >= 0; --j) v = v << 8 | h[j];
        return v;
    }
    uint32_t squeeze_challenge() { return (uint32_t)(squeeze_u64() % P); }            // from_bytes_mod_order, src/babybear.rs:65-71
    std::vector<uint32_t> squeeze_indices(size_t count, uint64_t mx) {                // src/transcript.rs:42-58
        std::vector<uint32_t> out;
        std::set<uint32_t> seen;
        while (out.size() < count) {
            const uint32_t idx = (uint32_t)(squeeze_u64() % mx);
            if (seen.insert(idx).second) out.push_back(idx);
        }
        return out;
    }

   private:
    std::vector<uint8_t> state_;
};

struct OpeningGroup {
    size_t tree_leaves;
    bool salted;
    std::vector<uint32_t> indices;
};

// StarkProof (src/fibonacci.rs:53-87) with the openings in the serialized record form the device writes
// (toyni_merkle_open_device); tests/harness/fib_prover.py::expand_proof turns the same layout into QueryProof structures.
struct Proof {
    size_t trace_len = 0, lde_size = 0;
    std::array<uint8_t, 32> trace_commitment{}, quotient_commitment{};
    uint32_t t_z = 0, t_gz = 0, t_ggz = 0, q_z = 0;
    std::vector<std::array<uint8_t, 32>> fri_commitments;
    std::vector<uint32_t> fri_final_layer;
    std::vector<uint32_t> query_indices;
    std::vector<uint8_t> opening_records;
    std::vector<OpeningGroup> opening_groups;
};

struct PhaseTimes {   // milliseconds, each phase closed by a stream synchronisation (only collected when asked for)
    double interpolate_lde_commit = 0, quotient_commit = 0, transcript_ood = 0, deep = 0, fri = 0, queries = 0;
};

class Prover {
   public:
    explicit Prover(size_t trace_len, int device = -1) : n_(trace_len), device_(device) {}
    ~Prover() { release(); }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;

    // Allocates every device buffer of a proof once (re-used by later proofs) and builds the two contexts.  "" = ok.
    std::string prepare() {
        if (ready_) return "";
        if (n_ < 8 || (n_ & (n_ - 1)) || n_ > ((size_t)1 << 22)) return "trace length must be a power of two in [8, 2^22]";
        log_n_ = 0;
        while (((size_t)1 << log_n_) < n_) ++log_n_;
        N_ = n_ << LOG_BLOWUP;
        log_N_ = log_n_ + LOG_BLOWUP;
        log_c_ = 0;                                                    // compact length 2^log_c >= n + MASK_DEGREE
        while (((size_t)1 << log_c_) < n_ + MASK_DEGREE) ++log_c_;
        size_t bound = 1;                                              // fri_degree_bound = next_power_of_two(n + MASK_DEGREE), :218
        while (bound < n_ + MASK_DEGREE) bound <<= 1;
        final_size_ = N_ / bound;
        if (final_size_ < 1 || log_c_ > log_N_) return "trace too short for this blow-up";
        sizes_.clear();
        for (size_t m = N_ / 2; m >= final_size_; m /= 2) { sizes_.push_back(m); if (m == final_size_) break; }
        salted_fri_ = 0;
        fri_digests_ = 0;
        fri_words_ = 0;
        for (size_t h : sizes_) { if (h != final_size_) salted_fri_ += h; fri_digests_ += toyni_merkle_total_digests(h); fri_words_ += h; }
        int st;
#define TOYNI_FIB_TRY(expr, what) do { st = (expr); if (st) return std::string(what) + ": " + toyni_error_string(st); } while (0)
        TOYNI_FIB_TRY(toyni_ntt_ctx_create((uint32_t)n_, device_, &ctx_n_), "trace-domain context");
        TOYNI_FIB_TRY(toyni_ntt_ctx_create((uint32_t)N_, device_, &ctx_N_), "LDE-domain context");
        TOYNI_FIB_TRY(toyni_stream_create(&stream_, toyni_ntt_ctx_device(ctx_N_)), "stream");
        TOYNI_FIB_TRY(toyni_stream_create(&side_, toyni_ntt_ctx_device(ctx_N_)), "stream");
        TOYNI_FIB_TRY(toyni_event_create(&salts_ready_), "event");
        const size_t tree_bytes = toyni_merkle_total_digests(N_) * 32;
        salt_bytes_ = ((3 * N_ + salted_fri_) * 16 + 63) & ~(size_t)63;
        struct { void** p; size_t bytes; } bufs[] = {
            {(void**)&d_compact_, ((size_t)4 << log_c_)}, {(void**)&d_trace_lde_, 4 * N_}, {(void**)&d_c_, 4 * N_}, {(void**)&d_q_, 4 * N_},
            {(void**)&d_qpoly_, 4 * N_}, {(void**)&d_deep_, 4 * N_}, {(void**)&d_layers_, 4 * fri_words_}, {(void**)&d_trace_tree_, tree_bytes},
            {(void**)&d_quot_tree_, tree_bytes}, {(void**)&d_deep_tree_, tree_bytes}, {(void**)&d_fri_trees_, fri_digests_ * 32},
            {(void**)&d_salts_, salt_bytes_}, {(void**)&d_ood_, 16}, {(void**)&d_idx_, 4 * max_openings()}, {(void**)&d_records_, max_record_bytes()}};
        for (auto& b : bufs) TOYNI_FIB_TRY(toyni_malloc(b.p, b.bytes), "device allocation");
        TOYNI_FIB_TRY(toyni_host_alloc((void**)&h_pinned_, pinned_bytes()), "pinned staging");
        ready_ = true;
        return "";
    }

    // trace: n canonical residues (the execution trace's one column, src/program/trace.rs).  key: 32 secret bytes (salts, mask).
    // Returns "" and fills `proof`, or the failing step.  "Constraint check at z failed" = the trace is not a Fibonacci column (:173-177).
    std::string generate_proof(const uint32_t* trace, const uint8_t key[32], Proof& proof, PhaseTimes* times = nullptr) {
        std::string err = prepare();
        if (!err.empty()) return err;
        int st;
        void* s = stream_;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](double& slot) {
            if (!times) return;
            (void)toyni_stream_synchronize(nullptr, s);
            const auto now = std::chrono::steady_clock::now();
            slot += std::chrono::duration<double, std::milli>(now - t_last).count();
            t_last = now;
        };
        const uint32_t g = root_of_unity(log_n_);
        uint32_t* hp = reinterpret_cast<uint32_t*>(h_pinned_);
        traffic_ = Traffic{};
        // every byte this proof moves over PCIe goes through these two (and the 32-byte roots the commit phase reads from pinned
        // memory): the totals are reported next to rocprofv3's --memory-copy-trace of the same program (profiles/r03_fib_prove_memcopy.txt)
        auto up = [&](void* d, const void* h, size_t bytes) { traffic_.h2d_bytes += bytes; ++traffic_.h2d_copies; return toyni_memcpy_h2d_async(d, h, bytes, s); };
        auto down = [&](void* h, const void* d, size_t bytes) { traffic_.d2h_bytes += bytes; ++traffic_.d2h_copies; return toyni_memcpy_d2h_async(h, d, bytes, s); };
        // every salt of the proof in one keystream: 3 LDE-size trees + the salted FRI layers (16 bytes per leaf, :341-343)
        // (on the side stream: the 134 MB keystream is generated while the trace is uploaded, interpolated and extended)
        TOYNI_FIB_TRY(toyni_chacha20_fill_device(d_salts_, salt_bytes_, key, 0, times ? s : side_), "salts");
        TOYNI_FIB_TRY(toyni_event_record(salts_ready_, times ? s : side_), "event");
        const uint8_t* salts_trace = d_salts_;
        const uint8_t* salts_quot = d_salts_ + 16 * N_;
        const uint8_t* salts_deep = d_salts_ + 32 * N_;
        const uint8_t* salts_fri = d_salts_ + 48 * N_;

        // ---- 1. trace polynomial + masking (:110-121): T_hat = T + (x^n - 1) R; LDE on the coset; commit ----
        TOYNI_FIB_TRY(toyni_memset_async(d_compact_, 0, (size_t)4 << log_c_, s), "clear");
        std::memcpy(hp, trace, 4 * n_);
        TOYNI_FIB_TRY(up(d_compact_, hp, 4 * n_), "trace upload");
        TOYNI_FIB_TRY(toyni_ntt_device(ctx_n_, d_compact_, d_compact_, 1, 1, s), "interpolation (INTT)");
        // masking on the host: only the coefficients it touches cross PCIe.  T_hat = T - R + x^n R touches [0, MASK_DEGREE) and
        // [n, n + MASK_DEGREE); the words above n are zero after the size-n INTT.  Short traces (n < MASK_DEGREE, e.g. the
        // reference's own trace_len 64) have the two ranges overlap: they are staged as one range [0, n + MASK_DEGREE).
        const bool overlap = n_ < MASK_DEGREE;
        const size_t nlow = overlap ? n_ : MASK_DEGREE;            // coefficients that can be non-zero among the first MASK_DEGREE
        uint32_t* stage = hp + n_;                                 // [0, MASK_DEGREE) then [n, n + MASK_DEGREE), or the one merged range
        const size_t stage_words = overlap ? n_ + MASK_DEGREE : 2 * (size_t)MASK_DEGREE;
        std::memset(stage, 0, 4 * stage_words);
        TOYNI_FIB_TRY(down(stage, d_compact_, 4 * nlow), "coefficients down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        {
            uint32_t ks[16];
            const uint32_t nonce[3] = {0u, 1u, 0u};   // nonce 1: the mask's keystream (nonce 0 is the salts')
            const auto kw = key_words(key);
            uint32_t* minus = stage;                                      // - R at x^i
            uint32_t* plus = overlap ? stage + n_ : stage + MASK_DEGREE;  // + R at x^(n + i)
            for (unsigned i = 0; i < MASK_DEGREE; ++i) {
                if (i % 8 == 0) chacha20_block(kw.data(), i / 8, nonce, ks);
                const uint64_t v = (uint64_t)ks[2 * (i % 8)] | (uint64_t)ks[2 * (i % 8) + 1] << 32;
                const uint32_t r = (uint32_t)(v % P);
                minus[i] = submod(minus[i], r);
                plus[i] = addmod(plus[i], r);
            }
        }
        if (overlap) {
            TOYNI_FIB_TRY(up(d_compact_, stage, 4 * (n_ + MASK_DEGREE)), "masked coefficients up");
        } else {
            TOYNI_FIB_TRY(up(d_compact_, stage, 4 * MASK_DEGREE), "masked coefficients up");
            TOYNI_FIB_TRY(up(d_compact_ + n_, stage + MASK_DEGREE, 4 * MASK_DEGREE), "masked coefficients up");
        }
        const size_t ncoef_t = n_ + MASK_DEGREE;       // trace_poly = compact[0 .. ncoef_t)
        TOYNI_FIB_TRY(toyni_lde_device(ctx_N_, d_compact_, d_trace_lde_, 1, log_N_ - log_c_, COSET_SHIFT, s), "LDE");
        // The trace tree is not needed before z is drawn (its root is absorbed together with the quotient tree's): it is built on a side
        // stream while the constraint / quotient kernels, the two INTTs and the quotient tree run on the main one.  A tree's upper ~14
        // levels are a chain of dependent hashes that keeps a handful of CUs busy; two trees side by side hide each other's chains.
        TOYNI_FIB_TRY(toyni_stream_wait(side_, s), "stream order");
        TOYNI_FIB_TRY(toyni_merkle_commit_device(d_trace_lde_, salts_trace, N_, d_trace_tree_, times ? s : side_), "trace commitment");
        if (times) lap(times->interpolate_lde_commit);

        // ---- 2. constraint & quotient (:133-153) and the reference's two ifft calls ----
        TOYNI_FIB_TRY(toyni_fib_quotient_device(ctx_N_, d_trace_lde_, d_c_, d_q_, LOG_BLOWUP, COSET_SHIFT, s), "constraint / quotient");
        TOYNI_FIB_TRY(toyni_coset_ntt_device(ctx_N_, d_c_, d_c_, 1, COSET_SHIFT, 1, s), "ifft (c_poly)");          // :145
        TOYNI_FIB_TRY(toyni_coset_ntt_device(ctx_N_, d_q_, d_qpoly_, 1, COSET_SHIFT, 1, s), "ifft (q_poly)");      // :151
        TOYNI_FIB_TRY(toyni_stream_wait_event(s, salts_ready_), "stream order");   // the main stream's first use of the salts
        TOYNI_FIB_TRY(toyni_merkle_commit_device(d_q_, salts_quot, N_, d_quot_tree_, s), "quotient commitment");
        TOYNI_FIB_TRY(toyni_stream_wait(s, side_), "stream order");          // both trees are complete before their roots are read
        const size_t root_off = (toyni_merkle_total_digests(N_) - 1) * 32;
        uint8_t* roots = reinterpret_cast<uint8_t*>(stage + stage_words);
        TOYNI_FIB_TRY(down(roots, d_trace_tree_ + root_off, 32), "root down");
        TOYNI_FIB_TRY(down(roots + 32, d_quot_tree_ + root_off, 32), "root down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        std::memcpy(proof.trace_commitment.data(), roots, 32);
        std::memcpy(proof.quotient_commitment.data(), roots + 32, 32);
        if (times) lap(times->quotient_commit);

        // ---- 3./4. Fiat-Shamir, out-of-domain evaluations (:155-183) ----
        Transcript tr;
        tr.absorb(proof.trace_commitment.data(), 32);
        tr.absorb(proof.quotient_commitment.data(), 32);
        uint32_t z;
        {   // derive_z_from_transcript, :379-399: z outside <w_N> and outside 7 <w_N> (then g z and g^2 z are too)
            const uint32_t inv7 = powmod(COSET_SHIFT, P - 2);
            do { z = tr.squeeze_challenge(); } while (powmod(z, N_) == 1 || powmod(mulmod(z, inv7), N_) == 1);
        }
        const uint32_t pts[3] = {z, mulmod(g, z), mulmod(mulmod(g, g), z)};
        TOYNI_FIB_TRY(toyni_poly_eval_device(ctx_N_, d_compact_, ncoef_t, pts, 3, d_ood_, s), "OOD evaluations of T");
        TOYNI_FIB_TRY(toyni_poly_eval_device(ctx_N_, d_qpoly_, N_, pts, 1, d_ood_ + 3, s), "OOD evaluation of Q");
        uint32_t* ood = reinterpret_cast<uint32_t*>(roots + 64);
        TOYNI_FIB_TRY(down(ood, d_ood_, 16), "OOD values down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        proof.t_z = ood[0]; proof.t_gz = ood[1]; proof.t_ggz = ood[2]; proof.q_z = ood[3];
        {
            const uint32_t c_z = mulmod(mulmod(submod(submod(proof.t_ggz, proof.t_gz), proof.t_z), submod(z, powmod(g, n_ - 1))), submod(z, powmod(g, n_ - 2)));
            if (c_z != mulmod(proof.q_z, submod(powmod(z, n_), 1))) return "Constraint check at z failed";   // :173-177
        }
        tr.absorb_field(proof.t_z); tr.absorb_field(proof.t_gz); tr.absorb_field(proof.t_ggz); tr.absorb_field(proof.q_z);
        if (times) lap(times->transcript_ood);

        // ---- 5. DEEP layer (:186-198) ----
        const uint32_t ood4[4] = {proof.t_z, proof.t_gz, proof.t_ggz, proof.q_z};
        TOYNI_FIB_TRY(toyni_fib_deep_device(ctx_N_, d_trace_lde_, d_q_, d_deep_, LOG_BLOWUP, COSET_SHIFT, z, ood4, s), "DEEP layer");
        if (times) lap(times->deep);

        // ---- 6. FRI (:200-247): layer 0's tree here, then the whole fold loop in one call with the transcript behind a callback ----
        TOYNI_FIB_TRY(toyni_merkle_commit_device(d_deep_, salts_deep, N_, d_deep_tree_, s), "DEEP commitment");
        TOYNI_FIB_TRY(down(roots, d_deep_tree_ + root_off, 32), "root down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        proof.fri_commitments.clear();
        proof.fri_commitments.emplace_back();
        std::memcpy(proof.fri_commitments[0].data(), roots, 32);
        tr.absorb(roots, 32);
        struct Cb { Transcript* tr; Proof* proof; } cb{&tr, &proof};
        unsigned rounds = 0;
        st = toyni_fri_commit_phase_device(
            ctx_N_, d_deep_, N_, COSET_SHIFT, final_size_, salted_fri_ ? salts_fri : nullptr,
            [](void* user, unsigned, const uint8_t* prev_root, uint32_t* beta_out) -> int {
                Cb* c = static_cast<Cb*>(user);
                if (prev_root) {                                   // absorb_commitment(root_k), :242-243
                    c->proof->fri_commitments.emplace_back();
                    std::memcpy(c->proof->fri_commitments.back().data(), prev_root, 32);
                    c->tr->absorb(prev_root, 32);
                }
                if (beta_out) *beta_out = c->tr->squeeze_challenge();   // beta_{k+1}, :223
                return 0;
            },
            &cb, d_layers_, d_fri_trees_, nullptr, &rounds, s);
        if (st) return std::string("FRI commit phase: ") + toyni_error_string(st);
        if (rounds != sizes_.size()) return "FRI round count mismatch";
        traffic_.d2h_bytes += 32 * (size_t)rounds;   // the per-round roots: written by the device into pinned host memory, no copy call
        traffic_.pinned_root_writes += rounds;
        proof.fri_final_layer.assign(final_size_, 0);
        TOYNI_FIB_TRY(down(ood, d_layers_ + (fri_words_ - final_size_), 4 * final_size_), "final layer down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        std::memcpy(proof.fri_final_layer.data(), ood, 4 * final_size_);
        if (times) lap(times->fri);

        // ---- 7. queries (:249-295): every opening gathered on the device, one copy down ----
        const size_t half0 = N_ / 2, B = (size_t)1 << LOG_BLOWUP;
        proof.query_indices = tr.squeeze_indices(NUM_QUERIES, half0);
        proof.opening_groups.clear();
        {
            OpeningGroup gt{N_, true, {}}, gq{N_, true, {}}, gd{N_, true, {}};
            for (uint32_t q : proof.query_indices) {
                gt.indices.push_back(q); gt.indices.push_back((uint32_t)((q + B) % N_)); gt.indices.push_back((uint32_t)((q + 2 * B) % N_));   // T(x), T(g x), T(g^2 x)
                gq.indices.push_back(q);
                gd.indices.push_back(q); gd.indices.push_back((uint32_t)(q + half0));                                                           // DEEP layer: qi and its pair
            }
            proof.opening_groups.push_back(gt); proof.opening_groups.push_back(gq); proof.opening_groups.push_back(gd);
            std::vector<uint32_t> cur = proof.query_indices;
            for (size_t li = 0; li + 1 < sizes_.size(); ++li) {          // folded layers except the final one (:270-283)
                const size_t half = sizes_[li] / 2;
                OpeningGroup gf{sizes_[li], true, {}};
                for (auto& c : cur) { c = (uint32_t)(c % half); gf.indices.push_back(c); gf.indices.push_back((uint32_t)(c + half)); }
                proof.opening_groups.push_back(gf);
            }
        }
        size_t nidx = 0, nbytes = 0;
        for (auto& gr : proof.opening_groups) { nidx += gr.indices.size(); nbytes += gr.indices.size() * toyni_merkle_open_record_bytes(gr.tree_leaves); }
        if (nidx > max_openings() || nbytes > max_record_bytes()) return "opening buffers too small";
        {
            size_t o = 0;
            for (auto& gr : proof.opening_groups) { std::memcpy(hp + o, gr.indices.data(), 4 * gr.indices.size()); o += gr.indices.size(); }
        }
        TOYNI_FIB_TRY(up(d_idx_, hp, 4 * nidx), "query positions up");
        {
            std::vector<toyni_merkle_open_group> og;
            size_t io = 0, bo = 0, lo = 0, dlo = 0, slo = 0;
            for (size_t k = 0; k < proof.opening_groups.size(); ++k) {
                auto& gr = proof.opening_groups[k];
                const uint8_t *levels, *salts;
                const uint32_t* values;
                if (k == 0) { levels = d_trace_tree_; values = d_trace_lde_; salts = salts_trace; }
                else if (k == 1) { levels = d_quot_tree_; values = d_q_; salts = salts_quot; }
                else if (k == 2) { levels = d_deep_tree_; values = d_deep_; salts = salts_deep; }
                else {
                    const size_t h = sizes_[k - 3];
                    levels = d_fri_trees_ + dlo * 32; values = d_layers_ + lo; salts = salts_fri + slo * 16;
                    lo += h; dlo += toyni_merkle_total_digests(h); slo += h;
                }
                og.push_back(toyni_merkle_open_group{levels, gr.tree_leaves, values, salts, d_idx_ + io, gr.indices.size(), d_records_ + bo});
                io += gr.indices.size();
                bo += gr.indices.size() * toyni_merkle_open_record_bytes(gr.tree_leaves);
            }
            TOYNI_FIB_TRY(toyni_merkle_open_groups_device(og.data(), og.size(), s), "openings");   // every tree's openings: one launch
        }
        proof.opening_records.resize(nbytes);
        TOYNI_FIB_TRY(down(h_pinned_, d_records_, nbytes), "opening records down");
        TOYNI_FIB_TRY(toyni_stream_synchronize(nullptr, s), "sync");
        std::memcpy(proof.opening_records.data(), h_pinned_, nbytes);
        if (times) lap(times->queries);
        proof.trace_len = n_;
        proof.lde_size = N_;
        return "";
#undef TOYNI_FIB_TRY
    }

    struct Traffic { size_t h2d_bytes = 0, d2h_bytes = 0, h2d_copies = 0, d2h_copies = 0, pinned_root_writes = 0; };
    const Traffic& traffic() const { return traffic_; }   // PCIe traffic of the last proof
    size_t folds() const { return sizes_.size(); }
    size_t final_layer_size() const { return final_size_; }

   private:
    static std::array<uint32_t, 8> key_words(const uint8_t key[32]) {
        std::array<uint32_t, 8> k;
        std::memcpy(k.data(), key, 32);
        return k;
    }
    size_t max_openings() const { return NUM_QUERIES * (3 + 1 + 2 + 2 * (sizes_.empty() ? 0 : sizes_.size() - 1)); }
    size_t max_record_bytes() const { return max_openings() * toyni_merkle_open_record_bytes(N_); }
    size_t pinned_bytes() const {
        size_t b = 4 * n_ + 4 * (n_ + 2 * MASK_DEGREE) + 64 + 16 + 4 * (final_size_ > 4 ? final_size_ : 4) + 64;
        if (b < max_record_bytes()) b = max_record_bytes();
        if (b < 4 * max_openings()) b = 4 * max_openings();
        return b;
    }
    void release() {
        for (void* p : {(void*)d_compact_, (void*)d_trace_lde_, (void*)d_c_, (void*)d_q_, (void*)d_qpoly_, (void*)d_deep_, (void*)d_layers_, (void*)d_trace_tree_,
                        (void*)d_quot_tree_, (void*)d_deep_tree_, (void*)d_fri_trees_, (void*)d_salts_, (void*)d_ood_, (void*)d_idx_, (void*)d_records_})
            if (p) (void)toyni_free(p);
        if (h_pinned_) (void)toyni_host_free(h_pinned_);
        if (stream_) (void)toyni_stream_destroy(stream_);
        if (side_) (void)toyni_stream_destroy(side_);
        if (salts_ready_) (void)toyni_event_destroy(salts_ready_);
        if (ctx_n_) (void)toyni_ntt_ctx_destroy(ctx_n_);
        if (ctx_N_) (void)toyni_ntt_ctx_destroy(ctx_N_);
    }

    size_t n_, N_ = 0, final_size_ = 0, salted_fri_ = 0, fri_digests_ = 0, fri_words_ = 0, salt_bytes_ = 0;
    unsigned log_n_ = 0, log_N_ = 0, log_c_ = 0;
    int device_;
    bool ready_ = false;
    Traffic traffic_;
    std::vector<size_t> sizes_;       // folded layer sizes N/2 ... final_size
    toyni_ntt_ctx *ctx_n_ = nullptr, *ctx_N_ = nullptr;
    void *stream_ = nullptr, *side_ = nullptr, *salts_ready_ = nullptr;
    uint32_t *d_compact_ = nullptr, *d_trace_lde_ = nullptr, *d_c_ = nullptr, *d_q_ = nullptr, *d_qpoly_ = nullptr, *d_deep_ = nullptr, *d_layers_ = nullptr,
             *d_ood_ = nullptr, *d_idx_ = nullptr;
    uint8_t *d_trace_tree_ = nullptr, *d_quot_tree_ = nullptr, *d_deep_tree_ = nullptr, *d_fri_trees_ = nullptr, *d_salts_ = nullptr, *d_records_ = nullptr;
    uint8_t* h_pinned_ = nullptr;
};

// In-field Fibonacci column (the reference test's u64 wrapping_add stops being a field sequence after fib(93))
inline std::vector<uint32_t> fibonacci_trace(size_t n) {
    std::vector<uint32_t> out(n);
    uint32_t a = 1, b = 1;
    for (size_t i = 0; i < n; ++i) { out[i] = a; const uint32_t c = addmod(a, b); a = b; b = c; }
    return out;
}

}  // namespace fib
}  // namespace toyni
