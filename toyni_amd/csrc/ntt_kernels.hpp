// Fused multi-stage BabyBear NTT passes for gfx950.
//
// What this replaces: the reference's per-stage launches -- ntt_kernel_bit_reverse
// (cuda/ntt_kernel.cu:103-113), log2(n) x ntt_kernel_butterfly (:119-137) and scale_by_inv_n
// (:140-143) -- i.e. (log2 n + 1) full HBM sweeps per transform.  Here a transform of size
// n = M_1 * ... * M_P (P <= 3, M_p <= 1024) is P sweeps; no bit-reversal pass exists (the digit
// reversal is absorbed in the last pass' store addressing) and n^-1 is folded into a twiddle table.
//
// Decomposition (decimation in frequency, natural order in and out; same values as src/ntt.rs:24-53):
//   input index  j = j_1 * n/M_1 + j_2 * n/(M_1 M_2) + ...      (j_1 most significant)
//   output index k = k_1 + M_1 k_2 + M_1 M_2 k_3 + ...           (k_1 least significant)
//   pass p, for every prefix (k_1..k_{p-1}) and every column j' < S_p = n/(M_1..M_p):
//     y_p[prefix][k_p][j'] = w_{L_p}^(j' k_p) * sum_{j_p} y_{p-1}[prefix][j_p][j'] * w_{M_p}^(j_p k_p),
//     L_p = M_p S_p.  The last pass has S_P = 1 (rows are contiguous) and scatters to natural order.
//
// One workgroup owns a tile of C columns (KIND_COL) or C rows (KIND_ROW_*) of one M-point
// sub-transform and runs it in two register steps:
//   step 1: each thread loads E1 = 2^LE1 elements straight from HBM into VGPRs (coalesced segments of
//           C or E2 consecutive words), runs the LE1 high-bit radix-2 Gentleman-Sande stages in
//           registers, and parks the tile in LDS;
//   step 2: each thread reads groups of E2 = 2^LE2 elements back (a different lane<->element map),
//           runs the LE2 low-bit stages with wave-uniform twiddles (scalar loads), applies the
//           inter-pass twiddle / scale and stores straight to HBM.
// So a pass touches HBM once each way and LDS once each way.  Passes with M <= 32 are single-step and
// use no LDS.  No MFMA: this is integer modular arithmetic (64 lanes x 32-bit VALU).
//
// The bodies take (block id, thread id, LDS pointer) explicitly and contain no HIP builtins, so
// tests/emu steps exactly this code on the CPU against the oracle.
#pragma once
#include "bb_field.hpp"

namespace toyni {

enum PassKind : int { KIND_COL = 0, KIND_ROW_T = 1, KIND_ROW_N = 2 };

struct PassArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* stage_tw;  // packed per-stage table of this pass' M: [2^t - 1 + x] = w_{2^(t+1)}^x, Montgomery form
    const uint32_t* tw_lo;     // KIND_COL: w_L^x,               x < 2^tw_lowbits  (Montgomery form)
    const uint32_t* tw_hi;     // KIND_COL: w_L^(y << tw_lowbits), y < L >> tw_lowbits
    uint32_t tw_lowbits;
    uint32_t log_S;            // KIND_COL: log2(columns per prefix block)
    uint32_t scale;            // Montgomery form of an extra factor applied by this pass (n^-1 of the inverse transform); 0 = none
    uint32_t log_n;            // KIND_ROW_T: log2 n
    uint32_t log_M1;           // KIND_ROW_T: log2 M_1 (k_1 range the tile rows run over)
    uint32_t log_mid;          // KIND_ROW_T: log2(n / (M_1 * M)) -- number of middle digits (1 for P = 2)
    uint64_t rows_total;       // KIND_ROW_N: number of rows (= batch); tiles may be ragged
};

// base + 32-bit BYTE offset: keeps the address math in 32 bits so that global loads/stores take the
// SGPR-base + VGPR-offset (+ immediate) form
TOYNI_HD uint32_t ld32(const uint32_t* base, uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off);
}
TOYNI_HD void st32(uint32_t* base, uint32_t byte_off, uint32_t v) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

constexpr uint32_t cx_bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
}

template <int KIND, int LE1, int LE2, int LC>
struct Pass {
    static_assert(LE2 <= LE1 && LE1 <= 5 && LE1 >= 1, "step sizes");
    static constexpr int LM = LE1 + LE2;
    static constexpr uint32_t M = 1u << LM, E1 = 1u << LE1, E2 = 1u << LE2, C = 1u << LC;
    static constexpr uint32_t T = C * E2;          // threads per workgroup
    static constexpr uint32_t G2 = E1 / E2;        // step-2 groups per thread
    static constexpr bool TWO_STEP = LE2 > 0;
    // LDS pitch of one tile row (KIND_ROW_*): breaks the power-of-two stride between rows
    static constexpr uint32_t PITCH = KIND == KIND_ROW_T ? M + (C <= 16 ? 2 : 1) : M + (E2 & 31u);
    static constexpr uint32_t LDS_WORDS = !TWO_STEP ? 0 : (KIND == KIND_COL ? M * C : C * PITCH);

    // LDS word of tile element (c, r): c = column (KIND_COL) or row (KIND_ROW_*), r = position in the sub-transform
    static TOYNI_HD uint32_t lds_word(uint32_t c, uint32_t r) {
        if (KIND == KIND_COL) {
            // [r][c]; with only 16 columns the bank is (r&1)*16 + c, so fold bit LE2 of r (lane-varying in
            // step 2, constant per instruction in step 1) into bit 0
            uint32_t rr = (C < 32) ? (r ^ ((r >> LE2) & 1u)) : r;
            return rr * C + c;
        } else {
            // [c][r] with the low digit rotated by the high digit: step-1 writes (lanes = low digit) and
            // step-2 reads (lanes = high digit) both spread over banks
            uint32_t hi = r >> LE2;
            uint32_t low = (r + hi) & (E2 - 1);
            return c * PITCH + (hi << LE2) + low;
        }
    }

    // A tile is addressed as (uniform 64-bit base pointer) + (32-bit per-thread element offset): every offset inside
    // a tile is < n <= 2^27, so the loads/stores use the SGPR-base + VGPR-offset form and no 64-bit VALU address math.
    struct Tile {
        const uint32_t* in;
        uint32_t* out;
        uint32_t in_cshift_or_stride;       // KIND_ROW_T: log2 of the row stride; others unused
        uint32_t col0;                      // KIND_COL: first column index j' of the tile
        uint32_t valid_c;                   // KIND_ROW_N: rows of this tile that exist
    };

    static TOYNI_HD Tile tile_of(const PassArgs& a, uint32_t bid) {
        Tile t;
        t.col0 = 0;
        t.valid_c = C;
        t.in_cshift_or_stride = 0;
        if (KIND == KIND_COL) {
            const uint32_t tiles_log = a.log_S - LC;
            const uint64_t prefix = (uint64_t)bid >> tiles_log;
            t.col0 = (bid & ((1u << tiles_log) - 1)) << LC;
            const uint64_t base = (prefix << (a.log_S + LM)) + t.col0;
            t.in = a.in + base;
            t.out = a.out + base;
        } else if (KIND == KIND_ROW_T) {
            const uint32_t mid = bid & ((1u << a.log_mid) - 1);
            const uint32_t k1_tiles_log = a.log_M1 - LC;
            const uint32_t k1_0 = ((bid >> a.log_mid) & ((1u << k1_tiles_log) - 1)) << LC;
            const uint64_t b = (uint64_t)bid >> (a.log_mid + k1_tiles_log);
            t.in_cshift_or_stride = a.log_n - a.log_M1;
            t.in = a.in + ((b << a.log_n) + ((uint64_t)k1_0 << t.in_cshift_or_stride) + ((uint64_t)mid << LM));
            t.out = a.out + ((b << a.log_n) + k1_0 + ((uint64_t)mid << a.log_M1));
        } else {
            const uint64_t row0 = (uint64_t)bid << LC;
            t.in = a.in + (row0 << LM);
            t.out = a.out + (row0 << LM);
            const uint64_t left = a.rows_total - row0;
            t.valid_c = left < C ? (uint32_t)left : C;
        }
        return t;
    }

    // element offset of tile element (c, r) in the input
    static TOYNI_HD uint32_t in_offset(const PassArgs& a, const Tile& t, uint32_t c, uint32_t r) {
        if (KIND == KIND_COL) return (r << a.log_S) + c;
        if (KIND == KIND_ROW_T) return (c << t.in_cshift_or_stride) + r;
        return (c << LM) + r;
    }
    // element offset of finished element (c, natural sub-index k) in the output
    static TOYNI_HD uint32_t out_offset(const PassArgs& a, uint32_t c, uint32_t k) {
        if (KIND == KIND_COL) return (k << a.log_S) + c;
        if (KIND == KIND_ROW_T) return c + (k << (a.log_n - LM));
        return (c << LM) + k;
    }

    // w_L^e (Montgomery form, canonical) from the pass boundary's two-level table
    static TOYNI_HD uint32_t boundary_tw(const PassArgs& a, uint32_t e) {
        const uint32_t lo = a.tw_lo[e & ((1u << a.tw_lowbits) - 1)];
        const uint32_t hi = a.tw_hi[e >> a.tw_lowbits];
        return mont_mul(hi, lo);
    }

    // Finish and store the NB = 2^LB elements a thread holds for one (c, khi): register i holds natural sub-index
    // k = (bitrev(i) << LSH) | khi.  KIND_COL multiplies by the inter-pass twiddle w_L^(j' k) = A * G^bitrev(i) with
    // A = w_L^(j' khi), G = w_L^(j' << LSH): two table lookups per thread and one running product, instead of a
    // gather per element (the inverse's n^-1, a.scale on the first pass, is folded into A).
    template <int LB, int LSH>
    static TOYNI_HD void finish(const PassArgs& a, const Tile& t, uint32_t c, uint32_t khi, uint32_t (&x)[1 << LB]) {
        constexpr uint32_t NB = 1u << LB;
        // out_offset is linear in k: element b sits at off0 + b * step (bytes)
        const uint32_t off0 = out_offset(a, c, khi) << 2;
        const uint32_t step = (out_offset(a, 0u, 1u << LSH) - out_offset(a, 0u, 0u)) << 2;
        if (KIND == KIND_COL) {
            const uint32_t jcol = t.col0 + c;
            uint32_t tw = boundary_tw(a, jcol * khi);
            if (a.scale) tw = mont_mul(tw, a.scale);
            const uint32_t g = boundary_tw(a, jcol << LSH);
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                st32(t.out, off0 + b * step, mont_mul(x[cx_bitrev(b, LB)], tw));
                if (b + 1 < NB) tw = mont_mul(tw, g);
            }
        } else {
            const bool scaled = KIND == KIND_ROW_N && a.scale != 0u;  // 1-pass inverse only (multi-pass: the first pass scales)
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                uint32_t v = x[cx_bitrev(b, LB)];
                if (scaled) v = mont_mul(v, a.scale);
                st32(t.out, off0 + b * step, v);
            }
        }
    }

    // LE bits of radix-2 DIF butterflies on x[0..2^LE), register index bit s <-> sub-transform bit s + shift.
    // `low` = the thread's bits below `shift` (0 when the stage twiddles are wave-uniform).
    template <int LE, int SHIFT>
    static TOYNI_HD void stages(uint32_t (&x)[1 << LE], const uint32_t* stage_tw, uint32_t low) {
#pragma unroll
        for (int s = LE - 1; s >= 0; --s) {
            const int d = 1 << s;
            const int t = s + SHIFT;  // butterfly spans 2^(t+1) elements
            const uint32_t* tw = stage_tw + ((1u << t) - 1u) + low;
            uint32_t w[1 << (LE - 1)];
#pragma unroll
            for (int q = 0; q < d; ++q) w[q] = tw[(uint32_t)q << SHIFT];
#pragma unroll
            for (int i = 0; i < (1 << LE); ++i) {
                if (i & d) continue;
                const uint32_t u = x[i], v = x[i + d];
                x[i] = bb_add(u, v);
                if (SHIFT == 0 && (i & (d - 1)) == 0) x[i + d] = bb_sub(u, v);  // twiddle w^0 = 1 (known at compile time)
                else x[i + d] = mont_mul(bb_sub_lazy(u, v), w[i & (d - 1)]);
            }
        }
    }

    static TOYNI_HD void phase1(const PassArgs& a, uint32_t bid, uint32_t tid, uint32_t* lds) {
        const Tile t = tile_of(a, bid);
        uint32_t c, lo;
        if (KIND == KIND_COL) { c = tid & (C - 1); lo = tid >> LC; }
        else { lo = tid & (E2 - 1); c = tid >> LE2; }
        const bool live = KIND != KIND_ROW_N || c < t.valid_c;  // only single-pass row tiles can be ragged

        uint32_t x[E1];
        // in_offset is linear in r: register i sits at off0 + i * step (bytes)
        const uint32_t off0 = in_offset(a, t, c, lo) << 2;
        const uint32_t step = (in_offset(a, t, 0u, E2) - in_offset(a, t, 0u, 0u)) << 2;
        if (live) {
#pragma unroll
            for (uint32_t i = 0; i < E1; ++i) x[i] = ld32(t.in, off0 + i * step);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < E1; ++i) x[i] = 0u;
        }

        stages<LE1, LE2>(x, a.stage_tw, lo);

        if (TWO_STEP) {
#pragma unroll
            for (uint32_t i = 0; i < E1; ++i) lds[lds_word(c, lo + (i << LE2))] = x[i];
        } else {
            if (live) finish<LE1, 0>(a, t, c, 0u, x);
        }
    }

    static TOYNI_HD void phase2(const PassArgs& a, uint32_t bid, uint32_t tid, const uint32_t* lds) {
        const Tile t = tile_of(a, bid);
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) {
            const uint32_t gamma = tid + g * T;
            uint32_t c, hi;
            if (KIND == KIND_ROW_N) { hi = gamma & (E1 - 1); c = gamma >> LE1; }
            else { c = gamma & (C - 1); hi = gamma >> LC; }

            uint32_t x[E2];
#pragma unroll
            for (uint32_t i = 0; i < E2; ++i) x[i] = lds[lds_word(c, (hi << LE2) + i)];

            stages<LE2, 0>(x, a.stage_tw, 0u);

            if (KIND != KIND_ROW_N || c < t.valid_c) finish<LE2, LE1>(a, t, c, bitrev32(hi, LE1), x);
        }
    }
};

// ---- u64 <-> u32 edge of the reference-shaped entry points (src/ntt.rs:233: &mut [BabyBear] as *mut u64) ----
// narrow also reduces mod p, so a non-canonical u64 behaves like BabyBear::new (src/babybear.rs:26-30)
TOYNI_HD uint32_t narrow_u64(uint64_t v) { return (uint32_t)(v % BB_P); }

// ---- FRI pairwise fold (src/math/fri.rs:27-48), structured-domain form ----
// Layer points are x_i = x0 * w_m^i (prover: src/fibonacci.rs:214,228-231), so
//   out[i] = (a+b)/2 + (a-b) * [ (beta / (2 x0)) * w_m^-i ],   a = evals[i], b = evals[i + m/2].
// w_m^-i = winv_N^(i << log_step) comes from the ctx' two-level inverse-root table.
struct FoldArgs {
    const uint32_t* evals;
    uint32_t* out;
    const uint32_t* inv_lo;   // winv_N^x,                 x < 2^lowbits (Montgomery)
    const uint32_t* inv_hi;   // winv_N^(y << lowbits)
    uint32_t lowbits;
    uint32_t log_step;        // log2(N / m)
    uint32_t coef;            // Montgomery form of beta / (2 x0)
    uint64_t half;            // m / 2
};

TOYNI_HD uint32_t fold_one(const FoldArgs& f, uint64_t i, uint32_t a, uint32_t b) {
    const uint32_t e = (uint32_t)(i << f.log_step);
    const uint32_t w = mont_mul(f.inv_hi[e >> f.lowbits], f.inv_lo[e & ((1u << f.lowbits) - 1)]);  // Montgomery form of w_m^-i
    const uint32_t cw = mont_mul(w, f.coef);                                                        // Montgomery form of coef * w_m^-i
    const uint32_t avg = bb_halve(bb_add(a, b));
    return bb_add(avg, mont_mul(bb_sub_lazy(a, b), cw));
}

// ---- FRI fold with explicit points (the reference's signature fri_fold(evals, xs, beta)) ----
// x^-1 by Fermat like src/babybear.rs:111-114, shared over BATCH elements with Montgomery's trick.
TOYNI_HD uint32_t bb_mul_plain(uint32_t a, uint32_t b) { return mont_mul(a, to_mont(b)); }
TOYNI_HD uint32_t bb_inv_dev(uint32_t a) {
    // a^(p-2), p-2 = 0x77FFFFFF
    const uint32_t aR = to_mont(a);
    uint32_t r = BB_R1;  // Montgomery one
    uint32_t base = aR;
    uint32_t e = BB_P - 2u;
    for (int i = 0; i < 31; ++i) {
        if (e & 1u) r = mont_mul(r, base);
        base = mont_mul(base, base);
        e >>= 1;
    }
    return from_mont(r);
}

}  // namespace toyni
