// Host-side plan of a size-n transform: how log2(n) is split into passes, and the twiddle tables
// each pass reads.  Replaces build_twiddles_device / ntt_ctx_create of the reference
// (cuda/ntt_kernel.cu:160-185, 213-234): instead of two packed tables of n-1 u64 each (2 GiB at
// n = 2^27) a context holds, per direction, one packed stage table per distinct pass size (< 1024
// words) and one two-level table per pass boundary (<= 24 K words), all u32 in Montgomery form.
// Plain C++ (shared with tests/emu).
#pragma once
#include <cstdlib>
#include <stdint.h>
#include <type_traits>
#include <vector>

#include "bb_field.hpp"
#include "ntt_kernels.hpp"

namespace toyni {

constexpr int MAX_PASSES = 3;
constexpr int MAX_LOG_N = 27;  // BabyBear two-adicity, src/babybear.rs:119-125

struct PassPlan {
    int kind;      // PassKind
    int log_m;     // log2 of this pass' sub-transform size
    int log_s;     // KIND_COL: log2(columns per prefix block) = log2(n / (M_1..M_p))
    // offsets (in u32 words) into the direction's table blob
    uint32_t stage_off, stage3_off;
    uint32_t lo_off, hi_off, lowbits;
};

struct NttPlan {
    int log_n = 0;
    int npasses = 0;
    PassPlan pass[MAX_PASSES];
    // domain table: two-level w_n^x (fwd blob) / w_n^-x (inv blob), x < n
    uint32_t dom_lo_off = 0, dom_hi_off = 0, dom_lowbits = 0;
    // compact low levels of its subgroups (SubDomain, ntt_kernels.hpp): dom_sub_off[s] -> w^(x << s), x < 2^(dom_lowbits - s), 0 < s < dom_lowbits
    uint32_t dom_sub_off[32] = {};
    uint32_t scale_inv = 0;             // Montgomery form of n^-1, applied by the first pass of an inverse transform
    // n = 2^13 .. 2^15: tables of the single-sweep LDS-resident kernel (LdsPass<lds_la>); lds_la = 0 otherwise
    int lds_la = 0;
    uint32_t lds_stage_a_off = 0, lds_stage_b_off = 0, lds_gtab_off = 0;
    // n = 2^11 / 2^12: the packed n-point stage table and its radix-4 companion (Row2048 / Row4096: one sweep, one / two waves per transform)
    uint32_t row_stage_off = 0, row_stage3_off = 0;
    std::vector<uint32_t> fwd, inv;     // table blobs, Montgomery form
};

// latency = true: the SECOND plan of n = 2^21 / 2^22, two passes with a 2048-point three-step pass.  Rounds 2-4: 4-wide latency tiles
// only -- one launch fewer for a lone transform -- while streaming launches kept the three-pass split below.  Round 5: the 2048-point
// pass also has a STREAMING shape (16-wide tiles, 32 elements per thread), so n = 2^21 runs this plan for launches of every size
// (has_stream2_plan), n = 2^22 for lone transforms and for its low-degree extensions (toyni_hip.hip: use_two_pass_plan)
// (n = 2^23 / 2^24 as 4096-point three-step passes were built and measured too: 54.6 against 50.7 us and 124 against 88 us for the
// three-pass plan -- a 4096 x 4 tile is one 1024-thread workgroup per CU with 16-byte row segments; not kept)
inline bool has_latency_plan(int log_n) { return log_n == 21 || log_n == 22; }
// Uneven splits: which pass gets the extra stage.  A column pass also applies the inter-pass twiddle and reads strided; the closing
// row pass does neither -- so the larger factor goes LAST (experiment switch TOYNI_SPLIT_SMALL_FIRST=0: first, as in round 1).
inline bool& split_small_first() {
    static bool v = [] { const char* env = std::getenv("TOYNI_SPLIT_SMALL_FIRST"); return env ? env[0] != '0' : true; }();
    return v;
}
// element `idx` of a comma-separated integer list in the environment (0 when absent)
inline int plan_env_triple(const char* name, int idx) {
    const char* env = std::getenv(name);
    if (!env) return 0;
    for (int i = 0; i < idx; ++i) {
        while (*env && *env != ',') ++env;
        if (!*env) return 0;
        ++env;
    }
    return std::atoi(env);
}
inline void split_passes(int log_n, int& npasses, int (&logm)[MAX_PASSES], bool latency = false) {
    // (the larger factor LAST, as in the two-pass plans below: at n = 2^21 the 2048-point pass is then the closing row pass, which has a
    // streaming form -- contiguous rows in, 64-byte segments out -- besides the 4-wide latency form)
    if (latency && has_latency_plan(log_n)) { npasses = 2; logm[0] = log_n / 2; logm[1] = (log_n + 1) / 2; logm[2] = 0; return; }
    if (log_n <= 10) { npasses = 1; logm[0] = log_n; logm[1] = logm[2] = 0; return; }
    const bool small_first = split_small_first();
    if (log_n <= 20) {
        npasses = 2;
        const int hi = (log_n + 1) / 2, lo = log_n / 2;
        const bool swap = small_first && lo >= 6;   // the smallest column shape is 64 points
        logm[0] = swap ? lo : hi;
        logm[1] = swap ? hi : lo;
        logm[2] = 0;
        return;
    }
    npasses = 3;
    const int a = (log_n + 2) / 3, b = (log_n + 1) / 3, c = log_n / 3;   // a >= b >= c
    logm[0] = small_first ? c : a;
    logm[1] = b;
    logm[2] = small_first ? a : c;
    // experiment switch TOYNI_SPLIT3="a,b,c": that split for the size a + b + c (column shapes exist for 6..10 stage bits, closing row
    // shapes for 5..10); the A/B of profiles/r05_ab_split3.txt
    static const int forced[3] = {plan_env_triple("TOYNI_SPLIT3", 0), plan_env_triple("TOYNI_SPLIT3", 1), plan_env_triple("TOYNI_SPLIT3", 2)};
    if (forced[0] + forced[1] + forced[2] == log_n && forced[0] >= 6 && forced[0] <= 10 && forced[1] >= 6 && forced[1] <= 10 && forced[2] >= 5 &&
        forced[2] <= 10) {
        logm[0] = forced[0]; logm[1] = forced[1]; logm[2] = forced[2];
    }
}

// packed stage table of size-M transform with root w_M: entry [2^t - 1 + x] = w_{2^(t+1)}^x, x < 2^t
inline void append_stage_table(std::vector<uint32_t>& blob, int log_m, uint32_t w_m) {
    for (int t = 0; t < log_m; ++t) {
        const uint32_t w_len = bb_pow_host(w_m, 1ull << (log_m - (t + 1)));
        uint32_t cur = 1;
        for (uint32_t x = 0; x < (1u << t); ++x) {
            blob.push_back(to_mont_host(cur));
            cur = bb_mul_host(cur, w_len);
        }
    }
}

// radix-4 companion of the stage table (Pass::stages_thread_r4): for t = 1 .. log_m - 1 and W = w_{2^(t+1)}, block t at offset
// 2^t - 2 holds W^(3x) for x < 2^(t-1), then -W^(3x + 2^(t-1)) for x < 2^(t-1).  2^log_m - 2 words.
inline void append_stage3_table(std::vector<uint32_t>& blob, int log_m, uint32_t w_m) {
    for (int t = 1; t < log_m; ++t) {
        const uint32_t w_len = bb_pow_host(w_m, 1ull << (log_m - (t + 1)));
        const uint32_t w3 = bb_mul_host(bb_mul_host(w_len, w_len), w_len);
        const uint32_t quarter = bb_pow_host(w_len, 1ull << (t - 1));
        uint32_t cur = 1;
        for (uint32_t x = 0; x < (1u << (t - 1)); ++x) {
            blob.push_back(to_mont_host(cur));
            cur = bb_mul_host(cur, w3);
        }
        cur = quarter;
        for (uint32_t x = 0; x < (1u << (t - 1)); ++x) {
            blob.push_back(to_mont_host(BB_P - cur));   // never 0: a power of a root of unity
            cur = bb_mul_host(cur, w3);
        }
    }
}

// two-level table of w (order 2^log_l): lo[x] = w^x (x < 2^lowbits), hi[y] = factor * w^(y << lowbits)
inline void append_two_level(std::vector<uint32_t>& blob, int log_l, uint32_t w, uint32_t factor,
                             uint32_t& lo_off, uint32_t& hi_off, uint32_t& lowbits) {
    lowbits = (uint32_t)((log_l + 1) / 2);
    lo_off = (uint32_t)blob.size();
    uint32_t cur = 1;
    for (uint32_t x = 0; x < (1u << lowbits); ++x) {
        blob.push_back(to_mont_host(cur));
        cur = bb_mul_host(cur, w);
    }
    hi_off = (uint32_t)blob.size();
    const uint32_t w_hi = cur;  // w^(2^lowbits)
    cur = factor;
    for (uint32_t y = 0; y < (1u << (log_l - lowbits)); ++y) {
        blob.push_back(to_mont_host(cur));
        cur = bb_mul_host(cur, w_hi);
    }
}

inline bool build_plan(int log_n, NttPlan& plan, bool latency = false) {
    if (log_n < 0 || log_n > MAX_LOG_N) return false;  // cuda/ntt_kernel.cu:217-220
    plan.log_n = log_n;
    int logm[MAX_PASSES];
    split_passes(log_n, plan.npasses, logm, latency);
    const uint32_t w_n = bb_root_of_unity_host((uint32_t)log_n);      // cuda/ntt_kernel.cu:222-223
    const uint32_t w_n_inv = bb_pow_host(w_n, (1ull << log_n) - 1);   // omega^(n-1), src/ntt.rs:59
    const uint32_t n_inv = bb_inv_host((uint32_t)((1ull << log_n) % BB_P));  // src/ntt.rs:62
    plan.scale_inv = to_mont_host(n_inv);

    for (int dir = 0; dir < 2; ++dir) {
        std::vector<uint32_t>& blob = dir ? plan.inv : plan.fwd;
        blob.clear();
        const uint32_t w = dir ? w_n_inv : w_n;
        int consumed = 0;
        for (int p = 0; p < plan.npasses; ++p) {
            PassPlan& pp = plan.pass[p];
            pp.log_m = logm[p];
            const bool last = p == plan.npasses - 1;
            pp.kind = plan.npasses == 1 ? KIND_ROW_N : (last ? KIND_ROW_T : KIND_COL);
            const int log_l = log_n - consumed;       // L_p
            pp.log_s = log_l - pp.log_m;
            // root of the sub-transform: w_n^(n / M)
            const uint32_t w_m = bb_pow_host(w, 1ull << (log_n - pp.log_m));
            const uint32_t stage_off = (uint32_t)blob.size();
            append_stage_table(blob, pp.log_m, w_m);
            const uint32_t stage3_off = (uint32_t)blob.size();
            append_stage3_table(blob, pp.log_m, w_m);
            uint32_t lo_off = 0, hi_off = 0, lowbits = 0;
            if (pp.kind == KIND_COL) {
                const uint32_t w_l = bb_pow_host(w, 1ull << (log_n - log_l));
                append_two_level(blob, log_l, w_l, 1u, lo_off, hi_off, lowbits);
            }
            if (dir == 0) { pp.stage_off = stage_off; pp.stage3_off = stage3_off; pp.lo_off = lo_off; pp.hi_off = hi_off; pp.lowbits = lowbits; }
            consumed += pp.log_m;
        }
        if (log_n >= 13 && log_n <= 15) {   // (n = 2^11 / 2^12 have the one- / two-waves-per-transform kernels instead, round 5: Row2048 / Row4096)
            plan.lds_la = log_n - 10;
            plan.lds_stage_a_off = (uint32_t)blob.size();
            append_stage_table(blob, plan.lds_la, bb_pow_host(w, 1ull << 10));            // w_{M_a} = w_n^1024
            plan.lds_stage_b_off = (uint32_t)blob.size();
            append_stage_table(blob, 10, bb_pow_host(w, 1ull << plan.lds_la));           // w_1024 = w_n^{M_a}
            plan.lds_gtab_off = (uint32_t)blob.size();
            uint32_t cur = 1;
            for (uint32_t j = 0; j < 1024; ++j) { blob.push_back(to_mont_host(cur)); cur = bb_mul_host(cur, w); }
        }
        if (log_n == 11 || log_n == 12) {
            plan.row_stage_off = (uint32_t)blob.size();
            append_stage_table(blob, log_n, w);
            plan.row_stage3_off = (uint32_t)blob.size();
            append_stage3_table(blob, log_n, w);
        }
        // domain table w_n^(+-x), x < n: the FRI fold reads the inverse one (x_i^-1 = x0^-1 * w_n^-i), the
        // multi-GPU 4-step transform both (same offsets in both blobs)
        append_two_level(blob, log_n, w, 1u, plan.dom_lo_off, plan.dom_hi_off, plan.dom_lowbits);
        for (uint32_t s = 1; s < plan.dom_lowbits; ++s) {   // the subgroups' compact low levels: 2^dom_lowbits words in all
            plan.dom_sub_off[s] = (uint32_t)blob.size();
            const uint32_t ws = bb_pow_host(w, 1ull << s);
            uint32_t cur = 1;
            for (uint32_t x = 0; x < (1u << (plan.dom_lowbits - s)); ++x) {
                blob.push_back(to_mont_host(cur));
                cur = bb_mul_host(cur, ws);
            }
        }
    }
    return true;
}

// the order-(n >> s) subgroup inside the plan's domain table, on the blob at `tables` (fwd or inv; host or device copy)
inline SubDomain sub_domain(const NttPlan& plan, const uint32_t* tables, int s) {
    const uint32_t L = plan.dom_lowbits;
    SubDomain d{};
    d.hi = tables + plan.dom_hi_off;
    if (s <= 0) { d.lo = tables + plan.dom_lo_off; d.lo_mask = (1u << L) - 1u; d.rsh = L; d.lsh = 0; }
    else if ((uint32_t)s < L) { d.lo = tables + plan.dom_sub_off[s]; d.lo_mask = (1u << (L - (uint32_t)s)) - 1u; d.rsh = L - (uint32_t)s; d.lsh = 0; }
    else { d.lo = tables + plan.dom_lo_off; d.lo_mask = 0; d.rsh = 0; d.lsh = (uint32_t)s - L; }   // lo[0] = 1
    return d;
}

// (kind, log_m) -> Pass<...> instantiation.  This table is the single place that fixes the step split and
// the tile width of every pass shape; f receives a value of the Pass type.
//
// The 1024-point passes come in three tile widths.  32 columns / rows = whole 128-byte lines per row segment (one
// 1024-thread workgroup per CU, 132 KiB of LDS): the streaming choice, +12 % over 16-wide on the batched workload.
// But a tile is 32 K elements, so a SINGLE 2^20 transform is only 32 tiles; when the launch has too few tiles to
// cover the chip the narrower variants (16 / 8 wide) are used -- the data is cache-resident at that size anyway.
// `log_tiles32` = log2 of the number of 32-wide tiles the launch would have.
// nt: the non-temporal variant of the same shape (streaming launches, see ld32 / st32).
// Launches of at most 2^pass3_max_log_tiles32() 32-wide tiles (a single transform, or a handful: the data is cache-resident
// and the chip is far from full) run the three-step shapes `Pass3` with 4-wide tiles instead: 8x the workgroups of the 32-wide
// shape and half the serial work per wave.  -1 = never.  (A variable so that tests/emu can step both executors and the library
// can take TOYNI_P3_TILES from the environment.)
// The three dispatch knobs below take their value from the environment ONCE, in their thread-safe static initialiser; the library
// never writes them afterwards (tests/emu does, to step both executors).
inline int plan_env_int(const char* name, int dflt) {
    const char* env = std::getenv(name);
    return env ? std::atoi(env) : dflt;
}
inline int& pass3_max_log_tiles32() {
    static int v = plan_env_int("TOYNI_P3_TILES", 6);   // -1: never the three-step shapes
    return v;
}

// launches that take the two-pass latency plan of n = 2^21 / 2^22 (has_latency_plan): those whose first pass has at most this
// many (log2) 32-wide tiles' worth of columns
inline int& lat_max_log_tiles32() {
    static int v = plan_env_int("TOYNI_LAT_TILES", 7);  // -1: never the two-pass latency plans
    return v;
}

// experiment switches (compile time): tile width (log2) of the wide variants of the 128-point and the 512-point passes
#ifndef TOYNI_WIDE_43
#define TOYNI_WIDE_43 6
#endif
#ifndef TOYNI_WIDE_54
#define TOYNI_WIDE_54 6
#endif
inline int& wide_min_log_tiles32() {
    static int v = plan_env_int("TOYNI_WIDE_TILES", 12);  // 99: never the 64-wide shapes
    return v;
}
// launches of at least this many (log2) 32-wide tiles run a 2048-point pass in its streaming three-step shape (16-wide tiles, 32
// elements per thread, ntt_pass3s_kernel) and n = 2^21 as the two sweeps of its "latency" plan; 99 = never (the three-pass plan)
inline int& stream3_min_log_tiles32() {
    static int v = plan_env_int("TOYNI_S3_TILES", 7);
    return v;
}
// Sizes whose two-pass plan beats the three-pass one on streaming launches.  n = 2^21: 1024-point column pass + 2048-point closing
// pass, 1.05 against 1.32 ms per 2^28 elements.  n = 2^22 as two 2048-point passes was built and measured too and is NOT taken
// (profiles/r05_ab_stream3.txt): 1.28-1.33 ms against 1.24-1.27 for 7/7/8 -- a 16-column tile means 64-byte row segments on both
// sides of the column pass, and with the inter-pass twiddle that pass takes 0.71 ms where the closing pass takes 0.57.  What the
// 16-wide column shape is kept for is the FIRST pass of a low-degree extension of n = 2^22 (it reads 2^-blow-up of its input:
// 0.95 against 1.21 ms for 64 x 2^17 -> 2^22), so only its zero-fraction variants are instantiated (dispatch_pass_lz).
inline bool has_stream2_plan(int log_n) { return log_n == 21; }

// LQ > 0: the interleaved (Ext, AoS) variants: the same table (a lone Ext vector counts as four transforms' worth of tiles), without
// the 8-wide two-step shapes and the 2048-point latency and column shapes (the streaming 2048-point CLOSING pass has its interleaved form).
template <int LQ = 0, class F>
inline bool dispatch_pass(int kind, int log_m, int log_tiles32, F&& f, bool nt = false) {
    // measured crossover (profiles/r02_latency.txt): 2^6 32-wide tiles for the 1024-point shapes, 2^7 for the 512-point and 2^8
    // for the 256-point ones (8 elements per thread: lighter, they win up to larger launches)
    // 2048-point passes exist only in the two-pass plans of n = 2^21 / 2^22 and only as three-step shapes (4-wide latency tiles here; the
    // 16-wide streaming closing pass above)
    // the streaming 2048-point closing pass (base and interleaved form; a lone Ext vector is already 2^7 32-wide tiles' worth at n = 2^21)
    if (log_m == 11 && kind == KIND_ROW_T && log_tiles32 >= stream3_min_log_tiles32()) {
        if (nt) f(Pass3<KIND_ROW_T, 5, 3, 3, 4, true, LQ>{}); else f(Pass3<KIND_ROW_T, 5, 3, 3, 4, false, LQ>{});
        return true;
    }
    if ((LQ == 0 && log_m == 11) || (pass3_max_log_tiles32() >= 0 && log_m >= 8 && log_m <= 10 && log_tiles32 <= pass3_max_log_tiles32() + (10 - log_m))) {
#define TOYNI_PASS3_CASE(K, A, B, D) if (kind == K && log_m == (A) + (B) + (D)) { f(Pass3<K, A, B, D, 2, false, LQ>{}); return true; }
        if constexpr (LQ == 0) {   // 2048-point passes exist only in the base form's latency plans of n = 2^21 / 2^22
            TOYNI_PASS3_CASE(KIND_COL, 4, 4, 3)
            TOYNI_PASS3_CASE(KIND_ROW_T, 4, 4, 3)
        }
        TOYNI_PASS3_CASE(KIND_COL, 3, 3, 2)
        TOYNI_PASS3_CASE(KIND_COL, 3, 3, 3)
        if constexpr (LQ == 0) {   // (a 1024-point FIRST pass exists at n = 2^20 only, where even one Ext vector is 128 32-wide tiles: beyond the threshold)
            TOYNI_PASS3_CASE(KIND_COL, 4, 3, 3)
        }
        TOYNI_PASS3_CASE(KIND_ROW_T, 3, 3, 2)
        TOYNI_PASS3_CASE(KIND_ROW_T, 3, 3, 3)
        TOYNI_PASS3_CASE(KIND_ROW_T, 4, 3, 3)
#undef TOYNI_PASS3_CASE
    }
#define TOYNI_PASS_GO(K, A, B, LC_) do { if (nt) f(Pass<K, A, B, LC_, true, LQ>{}); else f(Pass<K, A, B, LC_, false, LQ>{}); } while (0)
#define TOYNI_PASS_CASE(K, A, B, LC_) \
    if (kind == K && log_m == (A) + (B)) { TOYNI_PASS_GO(K, A, B, LC_); return true; }
#define TOYNI_PASS_CASE_W(K, A, B)                                                    \
    if (kind == K && log_m == (A) + (B)) {                                             \
        if (log_tiles32 >= 9) TOYNI_PASS_GO(K, A, B, 5);                               \
        else if (log_tiles32 >= 7 || LQ > 0) TOYNI_PASS_GO(K, A, B, 4);                \
        else { if constexpr (LQ == 0) TOYNI_PASS_GO(K, A, B, 3); }                     \
        return true;                                                                   \
    }
    // The 128-, 256- and 512-point passes (three-pass plans n >= 2^21, two-pass plans up to 2^19) are memory-bound, and the strided
    // pattern moves more with 256-byte row segments than with 128-byte ones (tools/membench.hip: 4.85 against 4.4 TB/s): large
    // launches take 64-wide tiles (wide_min_log_tiles32(): launches of at least that many 32-wide tiles; a huge value = never).
    // Measured (profiles/r02_ab_wide.txt): +7.7 % at 64 x 2^24, +6 % at 2^21 / 2^22, +5.4 % at 2^27, +3.8 % at 2^18; 128-wide tiles for
    // the 128-point passes add another 0.7-1.5 % at 2^21 / 2^22 and lose 0.3 % at 2^13 / 2^14 (not taken).  The 1024-point passes
    // of n = 2^20 cannot widen: a 64 x 1024 tile is 256 KiB.
#define TOYNI_PASS_CASE_WIDE(K, A, B, LCW)                                            \
    if (kind == K && log_m == (A) + (B)) {                                             \
        if (log_tiles32 >= wide_min_log_tiles32() + ((LCW) - 6)) TOYNI_PASS_GO(K, A, B, LCW); \
        else TOYNI_PASS_GO(K, A, B, 5);                                                \
        return true;                                                                   \
    }
    // strided column passes (first / middle passes of a 2- or 3-pass transform)
    TOYNI_PASS_CASE(KIND_COL, 3, 3, 5)
    TOYNI_PASS_CASE_WIDE(KIND_COL, 4, 3, TOYNI_WIDE_43)
    TOYNI_PASS_CASE_WIDE(KIND_COL, 4, 4, 6)
    TOYNI_PASS_CASE_WIDE(KIND_COL, 5, 4, TOYNI_WIDE_54)
    TOYNI_PASS_CASE_W(KIND_COL, 5, 5)
    // last pass of a multi-pass transform: contiguous rows in, transposed (natural order) out
    TOYNI_PASS_CASE(KIND_ROW_T, 5, 0, 6)
    TOYNI_PASS_CASE(KIND_ROW_T, 3, 3, 5)
    TOYNI_PASS_CASE_WIDE(KIND_ROW_T, 4, 3, TOYNI_WIDE_43)
    TOYNI_PASS_CASE_WIDE(KIND_ROW_T, 4, 4, 6)
    TOYNI_PASS_CASE_WIDE(KIND_ROW_T, 5, 4, TOYNI_WIDE_54)
    TOYNI_PASS_CASE_W(KIND_ROW_T, 5, 5)
    // single-pass transforms (n <= 1024): one row per batch entry.  Rows under 256 words never take the non-temporal kernels (the
    // launcher's rule, toyni_hip.hip: enqueue_transform), so those twins are not instantiated
#define TOYNI_PASS_CASE_PLAIN(K, A, B, LC_) \
    if (kind == K && log_m == (A) + (B)) { f(Pass<K, A, B, LC_, false, LQ>{}); return true; }
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 1, 0, 6)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 2, 0, 6)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 3, 0, 6)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 4, 0, 6)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 5, 0, 6)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 3, 3, 5)
    TOYNI_PASS_CASE_PLAIN(KIND_ROW_N, 4, 3, 5)
#undef TOYNI_PASS_CASE_PLAIN
    if constexpr (LQ == 0) {
        TOYNI_PASS_CASE(KIND_ROW_N, 4, 4, 4)
        TOYNI_PASS_CASE(KIND_ROW_N, 5, 4, 4)
        TOYNI_PASS_CASE(KIND_ROW_N, 5, 5, 3)
    } else {   // interleaved rows: 32 virtual rows = 8 Ext vectors per tile (>= 8 rows per coordinate: Pass::lds_row)
        TOYNI_PASS_CASE(KIND_ROW_N, 4, 4, 5)
        TOYNI_PASS_CASE(KIND_ROW_N, 5, 4, 5)
        TOYNI_PASS_CASE(KIND_ROW_N, 5, 5, 5)
    }
#undef TOYNI_PASS_CASE
#undef TOYNI_PASS_CASE_WIDE
#undef TOYNI_PASS_CASE_W
#undef TOYNI_PASS_GO
    return false;
}

// First pass of a low-degree extension (zero-padded input, PassArgs::in_prefix_log): the same column shapes with the
// zero fraction as a compile-time parameter, f(Pass{}, std::integral_constant<int, LZ>{}), LZ = 1..5.
template <int LQ = 0, class F>
inline bool dispatch_pass_lz(int log_m, int log_tiles32, int lz, F&& f) {
#define TOYNI_LZ_CASES(PASS)                                                                 \
    switch (lz) {                                                                            \
        case 1: f(PASS{}, std::integral_constant<int, 1>{}); return true;                    \
        case 2: f(PASS{}, std::integral_constant<int, 2>{}); return true;                    \
        case 3: f(PASS{}, std::integral_constant<int, 3>{}); return true;                    \
        case 4: f(PASS{}, std::integral_constant<int, 4>{}); return true;                    \
        case 5: f(PASS{}, std::integral_constant<int, 5>{}); return true;                    \
        default: return false;                                                               \
    }
#define TOYNI_COMMA ,
    // (the interleaved form too: the LDE of Ext vectors to 2^22 points -- a lone vector is 2^8 32-wide tiles' worth)
    if (log_m == 11 && log_tiles32 >= stream3_min_log_tiles32()) { TOYNI_LZ_CASES(Pass3<KIND_COL TOYNI_COMMA 5 TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 4 TOYNI_COMMA false TOYNI_COMMA LQ>) }
    if constexpr (LQ == 0) {
        if (log_m == 11) { TOYNI_LZ_CASES(Pass3<KIND_COL TOYNI_COMMA 4 TOYNI_COMMA 4 TOYNI_COMMA 3 TOYNI_COMMA 2>) }
    }
    if (pass3_max_log_tiles32() >= 0 && log_m >= 8 && log_m <= 10 && log_tiles32 <= pass3_max_log_tiles32() + (10 - log_m)) {
        if (log_m == 8) { TOYNI_LZ_CASES(Pass3<KIND_COL TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 2 TOYNI_COMMA 2 TOYNI_COMMA false TOYNI_COMMA LQ>) }
        if (log_m == 9) { TOYNI_LZ_CASES(Pass3<KIND_COL TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 2 TOYNI_COMMA false TOYNI_COMMA LQ>) }
        if constexpr (LQ == 0) {
            if (log_m == 10) { TOYNI_LZ_CASES(Pass3<KIND_COL TOYNI_COMMA 4 TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 2>) }
        }
    }
    if (log_m == 6) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 3 TOYNI_COMMA 3 TOYNI_COMMA 5 TOYNI_COMMA false TOYNI_COMMA LQ>) }
    if (log_m == 7) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 4 TOYNI_COMMA 3 TOYNI_COMMA 5 TOYNI_COMMA false TOYNI_COMMA LQ>) }
    if (log_m == 8) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 4 TOYNI_COMMA 4 TOYNI_COMMA 5 TOYNI_COMMA false TOYNI_COMMA LQ>) }
    if (log_m == 9) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 5 TOYNI_COMMA 4 TOYNI_COMMA 5 TOYNI_COMMA false TOYNI_COMMA LQ>) }
    if (log_m == 10) {
        if (log_tiles32 >= 9) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 5 TOYNI_COMMA 5 TOYNI_COMMA 5 TOYNI_COMMA false TOYNI_COMMA LQ>) }
        else if (log_tiles32 >= 7 || LQ > 0) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 5 TOYNI_COMMA 5 TOYNI_COMMA 4 TOYNI_COMMA false TOYNI_COMMA LQ>) }
        else { if constexpr (LQ == 0) { TOYNI_LZ_CASES(Pass<KIND_COL TOYNI_COMMA 5 TOYNI_COMMA 5 TOYNI_COMMA 3>) } }
    }
#undef TOYNI_COMMA
#undef TOYNI_LZ_CASES
    return false;
}

// Walks the passes of `batch` transforms src -> dst (src == dst allowed; `work` holds batch * n words and is
// needed when npasses > 1).  `tables` points at the direction's blob wherever the executor can read it
// (device memory for the HIP launcher, host memory for tests/emu).  launch(PassType{}, args, nblocks).
// n = 1 launches nothing: the caller copies src to dst if they differ.
// Coset scaling fused into the transform (BabyBearDomain, src/math/domain.rs:85-123,154-174): `s` is the factor whose
// powers scale the data (shift for a forward coset FFT, shift^-1 for an inverse one), lo/hi its two-level power table
// wherever the executor can read it.  lo == nullptr: plain transform.
struct CosetTables {
    const uint32_t* lo = nullptr;
    const uint32_t* hi = nullptr;
    uint32_t lowbits = 0;
    uint32_t s = 1;  // plain canonical
};

// lde_log > 0 (forward, multi-pass plans only): `src` holds the n >> lde_log leading words of every transform, the
// rest of the input is implied zeros -- the first pass runs its zero-aware variant (launch receives LZ = min(lde_log, 5)
// as std::integral_constant; 0 for every other launch).
// LQ > 0: `batch` counts base-field transforms, Q = 2^LQ of them interleaved word by word (batch = Q x the number of Ext vectors;
// src / work / dst hold batch * n words in the [vector][element][Q] layout): the interleaved variants of the same passes.
// The pieces layout around the exchange of a multi-device transform (PassArgs::in_split_* / out_split_*, toyni_ntt_slab_rows_device):
// the `batch` rows of this call (a rank's block of size-S_1 row transforms) live as [parts][batch][W], W = S_1 / parts, on the exchange
// side -- read by the FIRST pass of the forward row transforms, written (times w_N^-((row0 + b) kk)) by the LAST pass of the inverse ones.
// `unsupported` is set, and nothing more is launched, when the shape the dispatch table picks cannot address that layout (three-step
// shapes, fewer rows per piece than a thread's register stride, a single-pass plan); dry = only that check, no launch at all.
struct SlabIo {
    uint32_t log_parts = 0;
    const uint32_t* tw_lo = nullptr;   // big context's inverse domain table (inverse only)
    const uint32_t* tw_hi = nullptr;
    uint32_t tw_lowbits = 0, row0 = 0;
    bool dry = false;
    bool* unsupported = nullptr;
};

// the SLAB variant of a two-step shape (Pass<..., SLAB_ = true>); only the 32-wide plain shapes have one (see slab_variant_exists)
template <class P> struct SlabVariant { using type = P; };
template <int K, int A, int B, int C, bool N, int Q> struct SlabVariant<Pass<K, A, B, C, N, Q, false>> { using type = Pass<K, A, B, C, false, Q, true>; };
// The pieces layout is addressed by the 128-, 256- and 512-point column and closing shapes in their 32-wide, plain (not non-temporal)
// form: the row transforms of a multi-device transform are 2^14 .. 2^18 points (n = 2^21 .. 2^27, M1 = 2^7 .. 2^9), and one rank's
// launch is far below the footprint / tile counts from which the wide and non-temporal twins are picked.  Six kernels, not forty.
template <class P> constexpr bool slab_variant_exists() {
    if constexpr (P::STEPS == 2) return P::PASS_KIND != KIND_ROW_N && P::LQ == 0 && P::C == 32u && P::LM >= 7 && P::LM <= 9 && !P::NT;
    else return false;
}

template <int LQ = 0, class Launch>
inline bool for_each_pass(const NttPlan& plan, const uint32_t* tables, bool inverse, const uint32_t* src,
                          uint32_t* work, uint32_t* dst, uint64_t batch, Launch&& launch, const CosetTables& cs = CosetTables(),
                          int lde_log = 0, bool nt = false, const SlabIo* sio = nullptr) {
    if (plan.log_n == 0 || batch == 0) return true;  // n = 1: identity
    if (sio && (LQ != 0 || lde_log != 0 || cs.lo || plan.npasses < 2 || (batch & (batch - 1)) != 0 || (int)sio->log_parts >= plan.log_n)) return false;
    if (lde_log && (inverse || plan.npasses < 2 || lde_log > plan.pass[0].log_m)) return false;
    const uint64_t total_log = (uint64_t)plan.log_n;
    for (int p = 0; p < plan.npasses; ++p) {
        const PassPlan& pp = plan.pass[p];
        PassArgs a{};
        a.in = p == 0 ? src : work;
        a.out = p == plan.npasses - 1 ? dst : work;
        a.stage_tw = tables + pp.stage_off;
        a.stage_tw3 = tables + pp.stage3_off;
        a.tw_lo = tables + pp.lo_off;
        a.tw_hi = tables + pp.hi_off;
        a.tw_lowbits = pp.lowbits;
        a.log_S = (uint32_t)(pp.log_s + LQ);   // interleaved: Q words per column position (the twiddle column is column >> LQ)
        a.in_prefix_log = (uint32_t)(pp.log_s + pp.log_m - (p == 0 ? lde_log : 0) + LQ);
        a.nz_rows = (1u << pp.log_m) >> (p == 0 ? lde_log : 0);
        a.scale = (inverse && p == 0) ? plan.scale_inv : 0u;  // src/ntt.rs:62-65, fused
        a.log_n = (uint32_t)plan.log_n;
        a.log_M1 = (uint32_t)plan.pass[0].log_m;
        a.log_mid = (uint32_t)(plan.log_n - plan.pass[0].log_m - pp.log_m);
        a.rows_total = batch;
        // number of 32-wide tiles this launch would have (decides the tile width of the 1024-point passes)
        int log_tiles32 = 0;
        {
            const uint64_t tiles32 = (batch << (total_log - (uint64_t)pp.log_m)) >> 5;
            while ((2ull << log_tiles32) <= tiles32) ++log_tiles32;
        }
        auto body = [&](auto pass, auto lzc) {
            using P = decltype(pass);
            // forward coset FFT scales the INPUT of the first pass by s^j, inverse scales the OUTPUT of the last by s^k
            const bool cs_here = cs.lo && (inverse ? p == plan.npasses - 1 : p == 0);
            if (cs_here) {
                a.cs_lo = cs.lo;
                a.cs_hi = cs.hi;
                a.cs_lowbits = cs.lowbits;
                a.cs_mode = inverse ? 2u : 1u;
                // index distance between two consecutive registers of a thread
                uint64_t dj;
                if (!inverse) dj = pp.kind == KIND_COL ? (((uint64_t)1 << P::IN_STEP_LOG) << pp.log_s) : ((uint64_t)1 << P::IN_STEP_LOG);
                else dj = pp.kind == KIND_ROW_T ? ((uint64_t)1 << P::OUT_STEP_LOG) << (plan.log_n - P::LM)
                                                : ((uint64_t)1 << P::OUT_STEP_LOG);
                a.cs_g = to_mont_host(bb_pow_host(cs.s, dj));
            }
            uint64_t nblocks;
            if (pp.kind == KIND_ROW_N) nblocks = (batch + P::C - 1) / P::C;
            else nblocks = (batch << (total_log - P::LM)) / P::C;  // tiles of C columns / rows, each M long
            if (sio) {
                const uint32_t log_w = (uint32_t)plan.log_n - sio->log_parts;
                const uint32_t extra = (uint32_t)((batch - 1) << log_w);          // words between two pieces, beyond the W of one row
                const bool first = p == 0, last = p == plan.npasses - 1;
                bool ok_shape = true;              // (only the pass that touches the pieces is constrained: the others run as always)
                const bool special = (!inverse && first) || (inverse && last);
                if (!inverse && first) {           // forward: the first pass reads the pieces
                    ok_shape = slab_variant_exists<P>() && pp.kind == KIND_COL && pp.log_m >= (int)sio->log_parts + P::IN_STEP_LOG;
                    a.in_prefix_log = log_w;
                    a.in_split_shift = (uint32_t)pp.log_m - sio->log_parts;
                    a.in_split_extra = extra;
                }
                if (inverse && last) {             // inverse: the last pass writes them, twiddled
                    ok_shape = slab_variant_exists<P>() && pp.kind == KIND_ROW_T && P::LM >= (int)sio->log_parts + P::OUT_STEP_LOG;
                    a.out_prefix_log = log_w;
                    a.out_split_shift = (uint32_t)P::LM - sio->log_parts;
                    a.out_split_extra = extra;
                    a.cs_lo = sio->tw_lo;
                    a.cs_hi = sio->tw_hi;
                    a.cs_lowbits = sio->tw_lowbits;
                    a.cs_mode = 4u;
                    a.row0 = sio->row0;
                }
                if (batch == 1) ok_shape = false;  // (one row: extra = 0 would read as "contiguous"; the caller's fallback is exact)
                if (!ok_shape) { if (sio->unsupported) *sio->unsupported = true; return; }
                if (sio->dry || (sio->unsupported && *sio->unsupported)) return;
                if constexpr (slab_variant_exists<P>() && decltype(lzc)::value == 0) {
                    if (special) { launch(typename SlabVariant<P>::type{}, lzc, a, nblocks); return; }
                }
            }
            launch(pass, lzc, a, nblocks);
        };
        bool ok;
        // the pass that touches the pieces layout of a slab launch takes its 32-wide plain shape (the only ones with a SLAB variant)
        const bool slab_special = sio && ((!inverse && p == 0) || (inverse && p == plan.npasses - 1));
        if (slab_special && log_tiles32 >= wide_min_log_tiles32()) log_tiles32 = wide_min_log_tiles32() - 1;
        if (p == 0 && lde_log) ok = dispatch_pass_lz<LQ>(pp.log_m, log_tiles32, lde_log < 5 ? lde_log : 5, body);
        else ok = dispatch_pass<LQ>(pp.kind, pp.log_m, log_tiles32, [&](auto pass) { body(pass, std::integral_constant<int, 0>{}); }, nt && !slab_special);
        if (!ok) return false;
    }
    return true;
}

// Arguments of the single-sweep kernel for plans that have one (plan.lds_la != 0).  launch(LdsPass<LA>{}, args, ntiles).
template <class Launch>
inline bool lds_transform(const NttPlan& plan, const uint32_t* tables, bool inverse, const uint32_t* src, uint32_t* dst, uint64_t batch,
                          Launch&& launch, const CosetTables& cs = CosetTables(), int log_rows = 5) {
    if (!plan.lds_la) return false;
    if (batch == 0) return true;
    LdsArgs g{};
    g.in = src;
    g.out = dst;
    g.stage_a = tables + plan.lds_stage_a_off;
    g.stage_b = tables + plan.lds_stage_b_off;
    g.gtab = tables + plan.lds_gtab_off;
    g.scale = inverse ? plan.scale_inv : 0u;
    g.batch = batch;
    if (cs.lo) {
        g.cs_lo = cs.lo;
        g.cs_hi = cs.hi;
        g.cs_lowbits = cs.lowbits;
        g.cs_mode = inverse ? 2u : 1u;
        g.cs_g = to_mont_host(bb_pow_host(cs.s, inverse ? (32ull << plan.lds_la) : 1024ull));
    }
    // rows per workgroup: 2^log_rows (32 = one workgroup per CU; 16 / 8 = two / four decoupled ones), at least one transform
    if (log_rows > 5) log_rows = 5;
    if (log_rows < 3) log_rows = 3;
    if (log_rows < plan.lds_la) log_rows = plan.lds_la;
    const uint64_t per_tile = (1u << log_rows) >> plan.lds_la;
    const uint64_t ntiles = (batch + per_tile - 1) / per_tile;
#define TOYNI_LDS_CASE(A, B) if (plan.lds_la == A && log_rows == B) { launch(LdsPass<A, B>{}, g, ntiles); return true; }
    TOYNI_LDS_CASE(3, 5) TOYNI_LDS_CASE(4, 5) TOYNI_LDS_CASE(5, 5)
    TOYNI_LDS_CASE(3, 4) TOYNI_LDS_CASE(4, 4)
    TOYNI_LDS_CASE(3, 3)
#undef TOYNI_LDS_CASE
    return false;
}

// Arguments of the one-wave-per-transform kernel of n = 2^11 (Row2048, ntt_kernels.hpp) and of the two-waves-per-transform kernel of
// n = 2^12 (Row4096).  launch(args, rows).
template <class Launch>
inline bool row2048_transform(const NttPlan& plan, const uint32_t* tables, bool inverse, const uint32_t* src, uint32_t* dst, uint64_t batch,
                              Launch&& launch, const CosetTables& cs = CosetTables()) {
    if ((plan.log_n != 11 && plan.log_n != 12) || !plan.row_stage3_off) return false;
    if (batch == 0) return true;
    PassArgs a{};
    a.in = src;
    a.out = dst;
    a.stage_tw = tables + plan.row_stage_off;
    a.stage_tw3 = tables + plan.row_stage3_off;
    a.scale = inverse ? plan.scale_inv : 0u;
    a.rows_total = batch;
    if (cs.lo) {
        a.cs_lo = cs.lo;
        a.cs_hi = cs.hi;
        a.cs_lowbits = cs.lowbits;
        a.cs_mode = inverse ? 2u : 1u;
        // between two stores (k += n / 8) / two registers of step 1 (j += n / 32)
        a.cs_g = to_mont_host(bb_pow_host(cs.s, inverse ? (1ull << (plan.log_n - 3)) : (1ull << (plan.log_n - 5))));
    }
    launch(a, batch);
    return true;
}

// ---- one transform split over several devices (toyni_hip.h, section 2b) --------------------------------
// View x as [M_1][S_1] (M_1 = the plan's first-pass size): the first pass only couples elements of one column, so a
// rank that owns a contiguous block of columns runs it alone; what is left after it is, for every k_1, an ordinary
// size-S_1 transform over j' (src/ntt.rs:11-51 applied recursively).
//   forward : slab[j1][c] = x[j1 S_1 + col_base + c]  ->  w_n^((col_base + c) k1) * sum_j1 slab[j1][c] w_M1^(j1 k1), in place
//   inverse : the closing column transforms of the mirrored algorithm: sum_k1 slab[k1][c] w_M1^-(j1 k1) / M_1, no
//             twiddle (it is applied by the relayout before the exchange); `ones` = table of Montgomery ones,
//             max(2^lowbits, n >> lowbits) words, standing in for the inter-pass twiddle table.
// launch(PassType{}, args, nblocks).
template <class Launch>
inline bool slab_pass(const NttPlan& plan, const uint32_t* tables, bool inverse, uint32_t* slab, uint64_t cols_local,
                      uint64_t col_base, const uint32_t* ones, Launch&& launch) {
    if (plan.npasses < 2 || cols_local < 32 || (cols_local & (cols_local - 1))) return false;
    const PassPlan& pp = plan.pass[0];
    if (col_base + cols_local > (1ull << pp.log_s)) return false;
    int log_c = 0;
    while ((1ull << log_c) < cols_local) ++log_c;
    PassArgs a{};
    a.in = slab;
    a.out = slab;
    a.stage_tw = tables + pp.stage_off;
    a.stage_tw3 = tables + pp.stage3_off;
    a.tw_lowbits = pp.lowbits;
    a.log_S = (uint32_t)log_c;
    a.in_prefix_log = (uint32_t)(log_c + pp.log_m);
    a.nz_rows = 1u << pp.log_m;
    if (!inverse) {
        a.tw_lo = tables + pp.lo_off;
        a.tw_hi = tables + pp.hi_off;
        a.col_base = (uint32_t)col_base;
    } else {
        a.tw_lo = ones;
        a.tw_hi = ones;
        a.scale = to_mont_host(bb_inv_host((uint32_t)(1u << pp.log_m)));
    }
    int log_tiles32 = log_c - 5;
    if (log_c < 6 && log_tiles32 >= wide_min_log_tiles32()) log_tiles32 = wide_min_log_tiles32() - 1;  // a 32-column slab cannot hold a 64-wide tile
    return dispatch_pass(KIND_COL, pp.log_m, log_tiles32, [&](auto pass) {
        using P = decltype(pass);
        launch(pass, a, cols_local / P::C);
    });
}

}  // namespace toyni
