// Fused multi-stage BabyBear NTT passes for gfx950.
//
// What this replaces: the reference's per-stage launches -- ntt_kernel_bit_reverse
// (cuda/ntt_kernel.cu:103-113), log2(n) x ntt_kernel_butterfly (:119-137) and scale_by_inv_n
// (:140-143) -- i.e. (log2 n + 1) full HBM sweeps per transform.  Here a transform of size
// n = M_1 * ... * M_P (P <= 3, M_p <= 1024) is P sweeps; no bit-reversal pass exists (the digit
// reversal is absorbed in the last pass' store addressing) and n^-1 rides on the first pass' twiddle seed.
//
// Decomposition (decimation in frequency, natural order in and out; same values as src/ntt.rs:24-53):
//   input index  j = j_1 * n/M_1 + j_2 * n/(M_1 M_2) + ...      (j_1 most significant)
//   output index k = k_1 + M_1 k_2 + M_1 M_2 k_3 + ...           (k_1 least significant)
//   pass p, for every prefix (k_1..k_{p-1}) and every column j' < S_p = n/(M_1..M_p):
//     y_p[prefix][k_p][j'] = w_{L_p}^(j' k_p) * sum_{j_p} y_{p-1}[prefix][j_p][j'] * w_{M_p}^(j_p k_p),
//     L_p = M_p S_p.  The last pass has S_P = 1 (rows are contiguous) and scatters to natural order.
//
// One workgroup owns tiles of C columns (KIND_COL) or C rows (KIND_ROW_*) of one M-point sub-transform; it is
// persistent (loops over tiles, the next tile's HBM loads are in flight while the current one is finished) and
// runs a tile in two register steps:
//   step 1: each thread loads E1 = 2^LE1 elements straight from HBM into VGPRs (coalesced segments of
//           C or E2 consecutive words), runs the LE1 high-bit radix-2 Gentleman-Sande stages in
//           registers, and parks the tile in LDS;
//   step 2: each thread reads groups of E2 = 2^LE2 elements back (a different lane<->element map),
//           runs the LE2 low-bit stages with wave-uniform twiddles (held in SGPRs), applies the
//           inter-pass twiddle / scale and stores straight to HBM.
// So a pass touches HBM once each way and LDS once each way.  Passes with M <= 32 are single-step and
// use no LDS.  No MFMA: this is integer modular arithmetic (64 lanes x 32-bit VALU).
//
// The bodies take (tile id, thread id, LDS pointer) explicitly; the few device-only hooks (scheduling fences, wait
// counts, readfirstlane) are macros that vanish on the host, so tests/emu steps exactly this code on the CPU against
// the oracle (also under ASan/UBSan).
#pragma once
#include "bb_field.hpp"

namespace toyni {

enum PassKind : int { KIND_COL = 0, KIND_ROW_T = 1, KIND_ROW_N = 2 };

struct PassArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* stage_tw;  // packed per-stage table of this pass' M: [2^t - 1 + x] = w_{2^(t+1)}^x, Montgomery form
    const uint32_t* stage_tw3; // radix-4 companion: block t (t >= 1) at [2^t - 2]: w_{2^(t+1)}^(3x), then -w_{2^(t+1)}^(3x + 2^(t-1)), x < 2^(t-1)
    const uint32_t* tw_lo;     // KIND_COL: w_L^x,               x < 2^tw_lowbits  (Montgomery form)
    const uint32_t* tw_hi;     // KIND_COL: w_L^(y << tw_lowbits), y < L >> tw_lowbits
    uint32_t tw_lowbits;
    uint32_t log_S;            // KIND_COL: log2(columns per prefix block)
    uint32_t scale;            // Montgomery form of an extra factor applied by this pass (n^-1 of the inverse transform); 0 = none
    uint32_t log_n;            // KIND_ROW_T: log2 n
    uint32_t log_M1;           // KIND_ROW_T: log2 M_1 (k_1 range the tile rows run over)
    uint32_t log_mid;          // KIND_ROW_T: log2(n / (M_1 * M)) -- number of middle digits (1 for P = 2)
    uint64_t rows_total;       // KIND_ROW_N: number of rows (= batch); tiles may be ragged
    // Fused coset scaling of BabyBearDomain (src/math/domain.rs:154-174), replacing the host's serial shift^i loop:
    // cs_mode 1 = multiply INPUT element j of every transform by s^j (first pass of a forward coset FFT),
    // cs_mode 2 = multiply OUTPUT element k by s^k (last pass of an inverse one, s = shift^-1); 0 = off.
    // cs_lo/cs_hi: two-level table of s^x, x < n (Montgomery); cs_g: Montgomery form of the power of s between two
    // consecutive registers of a thread (uniform per launch, computed by the host for the pass shape).
    const uint32_t* cs_lo;
    const uint32_t* cs_hi;
    uint32_t cs_lowbits;
    uint32_t cs_mode;
    uint32_t cs_g;
    // KIND_COL on a column SLAB of the [M_1][S] view (one rank of a multi-GPU transform, toyni_ntt_slab_pass_device):
    // local column c is global column col_base + c -- only the inter-pass twiddle sees the difference.
    uint32_t col_base;
    // KIND_COL input addressing: log2 of the distance (words) between two prefix blocks of the INPUT.  Ordinarily
    // log_S + log2 M (same as the output); smaller for the first pass of a low-degree extension, whose input holds only
    // the n >> LZ leading (nonzero) words of every transform (toyni_lde_device).  nz_rows = input rows that exist.
    uint32_t in_prefix_log;
    uint32_t nz_rows;
    // ---- the pieces layout around the exchange of a multi-device transform (toyni_ntt_slab_rows_device, round 5) ----
    // A rank's block of rows (size-S_1 transforms over j') travels as [parts][rows][W], W = S_1 / parts: piece g holds the columns
    // j' in [g W, (g + 1) W) of every row.  Instead of a relayout sweep on either side of the exchange, the row transforms address it
    // directly:
    //   KIND_COL, first pass of the forward row transforms: input row r of a tile belongs to piece r >> in_split_shift, and every
    //     piece index adds in_split_extra WORDS to the linear address (in_prefix_log = log2 W); 0 = contiguous rows
    //   KIND_ROW_T, last pass of the inverse row transforms: output sub-index k of a tile belongs to piece k >> out_split_shift,
    //     out_split_extra words per piece index, transforms 2^out_prefix_log words apart (0 = log_n: contiguous rows); and with
    //     cs_mode = 4 the output kk of row b is multiplied by w_N^-((row0 + b) kk) -- the twiddle between the row transforms and the
    //     closing column transforms of the mirrored algorithm -- from the big context's inverse domain table in cs_lo / cs_hi.
    uint32_t in_split_shift, in_split_extra;
    uint32_t out_split_shift, out_split_extra, out_prefix_log;
    uint32_t row0;
};

// Diagnostic builds only (-DTOYNI_ABLATE=1|2|3, never the shipped library): bit 0 replaces tile loads by register
// arithmetic, bit 1 makes tile stores conditional on a value that never occurs -- timing-only builds that price the
// HBM side of a pass against its VALU/LDS side (results are garbage by construction).
#ifndef TOYNI_ABLATE
#define TOYNI_ABLATE 0
#endif

// base + 32-bit BYTE offset: keeps the address math in 32 bits so that global loads/stores take the
// SGPR-base + VGPR-offset (+ immediate) form
// NT: non-temporal hint.  The batched workload streams data that is read once and written once per pass; marking those
// accesses non-temporal is worth +2-3 % there (+9 % on batched 2^24) but costs cache-resident launches their reuse, so the
// launcher picks the variant by footprint (TOYNI_NT_MIN_BYTES).
template <bool NT = false>
TOYNI_HD uint32_t ld32(const uint32_t* base, uint32_t byte_off) {
#if TOYNI_ABLATE & 1
    return (uint32_t)(reinterpret_cast<uintptr_t>(base) >> 2) + byte_off;
#elif defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off));
    else return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off);
#else
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off);
#endif
}
template <bool NT = false>
TOYNI_HD void st32(uint32_t* base, uint32_t byte_off, uint32_t v) {
#if TOYNI_ABLATE & 2
    if (v == 0xFFFFFFFFu)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off));
    else
#endif
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// 16-byte accesses for threads that own CONSECUTIVE words (the single-step shapes: one thread per row of <= 32 points).  The type is
// declared 4-byte aligned: the rows of a caller's buffer need no more than that, and gfx950 takes unaligned dwordx4 accesses.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#endif

// Buffer-resource form of the same two accesses: address = resource base (SGPRs) + per-thread byte offset (one VGPR, constant
// over the registers of a tile) + uniform byte offset (an SGPR) -- no VALU instruction per access, where the pointer form costs one
// 64-bit VALU add per register (the compiler turns base + i * step into a running VGPR address: 108 v_lshl_add_u64 per tile of the
// 1024-point column pass, 8 % of its VALU issue).  Offsets are tile-relative (< 2^31).  Measured, alternating A/B against the pointer
// form (profiles/r02_ab_buffer.txt): stores +0.7 % at 1024 x 2^20, +1.6 % at 4096 x 2^18; loads as well +1.3-2.7 % at 64 x 2^24 and
// +1.5-2 % at 2^18, but -5.5 % in the 32 x 32 shapes of n = 2^20 (their 32 prefetch loads per thread: the pointer form stays there).
// -DTOYNI_NO_BUFFER_ADDR: pointer form everywhere (A/B builds).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TOYNI_NO_BUFFER_ADDR)
#define TOYNI_BUF 1
using BufRsrc = __amdgpu_buffer_rsrc_t;
TOYNI_HD BufRsrc buf_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000); }
template <bool NT = false>
TOYNI_HD uint32_t ldb32(BufRsrc r, uint32_t voff, uint32_t soff) {
#if TOYNI_ABLATE & 1
    return voff + soff;
#else
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, NT ? 2 : 0);
#endif
}
template <bool NT = false>
TOYNI_HD void stb32(BufRsrc r, uint32_t voff, uint32_t soff, uint32_t v) {
#if TOYNI_ABLATE & 2
    if (v == 0xFFFFFFFFu)
#endif
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, NT ? 2 : 0);
}
#else
#define TOYNI_BUF 0
#endif

constexpr uint32_t cx_bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
}

// LQ_ > 0 (round 4): Q = 2^LQ_ independent transforms INTERLEAVED word by word -- element j of transform q sits at word
// (j << LQ) + q -- which is the AoS layout of an Ext vector (4 x u32 per element, src/ext.rs / src/math/domain.rs:129-151: an
// Ext transform is the base transform of each coordinate).  A column pass sees such data as Q times as many columns whose twiddle
// column is (column >> LQ) (the host passes log_S / in_prefix_log inflated by LQ); the row kinds run C "virtual rows"
// c = (row << LQ) | q per tile, lanes running over (q, position) so that loads and stores stay contiguous, and park virtual row c
// in LDS row rho(c) (coordinate-major within blocks of 8 rows: conflict-free for both the step-1 writes and the step-2 reads).
// SLAB_ (round 5): the variant of a shape that addresses the PIECES layout of a multi-device transform (PassArgs::in_split_* /
// out_split_*, toyni_ntt_slab_rows_device).  A compile-time variant, not a run-time branch: an alternating A/B of the round-4 library
// against one with those branches inside every kernel read +0.5 % at 1024 x 2^20 and +3 % at 4096 x 2^18 (profiles/r05_ab_lib.txt) --
// the ordinary kernels must not know the layout exists.
template <int KIND, int LE1, int LE2, int LC, bool NT_ = false, int LQ_ = 0, bool SLAB_ = false>
struct Pass {
    static constexpr bool SLAB = SLAB_;
    static constexpr int PASS_KIND = KIND;
    static_assert(!SLAB_ || ((KIND == KIND_COL || KIND == KIND_ROW_T) && LQ_ == 0 && LE2 > 0), "pieces layout: two-step column / closing passes of base-field rows");
    static_assert(LE2 <= LE1 && LE1 <= 5 && LE1 >= 1, "step sizes");
    static constexpr int LQ = LQ_;
    static constexpr uint32_t Q = 1u << LQ_;
    static_assert(LQ_ == 0 || LQ_ == 2, "interleave: none, or the four coordinates of an Ext element");
    static_assert(LQ_ == 0 || KIND == KIND_COL || LC - LQ_ >= (KIND == KIND_ROW_T ? 2 : 3) || LE2 == 0, "interleaved rows: 4 or >= 8 rows per coordinate");
    static constexpr int LM = LE1 + LE2;
    static constexpr int LE1_ = LE1;
    static constexpr int STEPS = LE2 > 0 ? 2 : 1;
    // log2 of the row distance between two consecutive registers of a thread on the way in (step 1) and of the sub-index
    // distance between two consecutive stores on the way out (last step): what the host needs for the coset factor cs_g
    static constexpr int IN_STEP_LOG = LE2, OUT_STEP_LOG = LE2 > 0 ? LE1 : 0;
    static constexpr bool NT = NT_;
    static constexpr uint32_t M = 1u << LM, E1 = 1u << LE1, E2 = 1u << LE2, C = 1u << LC;
    static constexpr uint32_t T = C * E2;          // threads per workgroup
    static constexpr uint32_t G2 = E1 / E2;        // step-2 groups per thread
    static constexpr bool TWO_STEP = LE2 > 0;

    // ---- LDS tile layout -------------------------------------------------------------------------------
    // Element (c, r) with r = hi * E2 + low.  Both layouts are LINEAR in (c, hi, low), so every LDS access of a
    // thread is (one base register) + (compile-time immediate), and the pads make step-1 writes (lanes = c/low)
    // and step-2 reads (lanes = c/hi) conflict-free in the 32-lane groups ds_read_b32 / ds_write_b32 are served in:
    //   KIND_COL  : word = hi * (E2*C + PADC) + low * C + c       PADC = 16 when C = 16: bank = ((hi+low)&1)*16 + c
    //   KIND_ROW_*: word = c * PITCH + hi * (E2 + 1) + low        hi stride 33 = 1 mod 32; PITCH spreads the rows
    static constexpr uint32_t PADC = C < 32 ? 32 - C : 0;
    static constexpr uint32_t HI_STRIDE = KIND == KIND_COL ? E2 * C + PADC : E2 + 1;
    static constexpr uint32_t LOW_STRIDE = KIND == KIND_COL ? C : 1;
    static constexpr uint32_t row_pitch() {
        uint32_t base = E1 * (E2 + 1);
        // residue of PITCH mod 32 that spreads the tile rows a 32-lane group touches in step 2
        uint32_t want = KIND == KIND_ROW_T ? (C <= 16 ? 2u : 1u) : (LQ_ > 0 ? 1u : (E1 < 32 ? E1 : 0u));
        return base + ((want + 32u - (base & 31u)) & 31u);
    }
    static constexpr uint32_t PITCH = row_pitch();
    static constexpr uint32_t C_STRIDE = KIND == KIND_COL ? 1 : PITCH;
    static constexpr uint32_t LDS_WORDS = !TWO_STEP ? 0 : (KIND == KIND_COL ? E1 * HI_STRIDE : C * PITCH);

    // occupancy target: workgroups per CU by LDS (160 KiB), expressed as waves per SIMD for __launch_bounds__
    // (capped at 4: with the prefetch pipeline 16 waves per CU hide the rest, and 128 VGPRs are needed)
    static constexpr uint32_t lds_wg_per_cu() {
        return LDS_WORDS ? (160u * 1024u) / ((LDS_WORDS + (TWO_STEP ? M - E2 : 0u) + ((TWO_STEP && LE1 >= 2 && LC >= 4) ? M - 2u * E2 : 0u)) * 4u) : 8u;  // + the twiddle slices
    }
    static constexpr uint32_t min_waves_per_simd() {
        uint32_t w = (lds_wg_per_cu() * (T / 64u ? T / 64u : 1u) + 3u) / 4u;
        return w < 1u ? 1u : (w > 4u ? 4u : w);
    }
    static constexpr uint32_t MIN_WAVES = min_waves_per_simd();

    // LDS row of virtual row c = (cc << LQ) | q of an interleaved row tile.  >= 8 rows per coordinate: 8 q + (cc & 7) + 32 (cc >> 3)
    // with PITCH = 1 mod 32 -- a step-1 wave (lanes over q and eight consecutive positions) hits banks 8 q + position, a step-2
    // group of 32 consecutive virtual rows 32 different rows mod 32; 4 rows per coordinate (16-row tiles): 4 q + cc, PITCH = 2 mod 32.
    static TOYNI_HD uint32_t lds_row(uint32_t c) {
        if (LQ_ == 0 || KIND == KIND_COL) return c;
        const uint32_t q = c & (Q - 1u), cc = c >> LQ_;
        if ((C >> LQ_) >= 8u) return (q << 3) + (cc & 7u) + ((cc >> 3) << 5);
        return q * (C >> LQ_) + cc;
    }
    static TOYNI_HD uint32_t lds_word(uint32_t c, uint32_t hi, uint32_t low) { return lds_row(c) * C_STRIDE + hi * HI_STRIDE + low * LOW_STRIDE; }

    // ---- tiles -----------------------------------------------------------------------------------------
    // A tile is addressed as (uniform 64-bit base pointer) + (32-bit per-thread BYTE offset): every offset inside a
    // tile is < 4n <= 2^29, so loads/stores take the SGPR-base + VGPR-offset form and no 64-bit VALU address math.
    struct Tile {
        const uint32_t* in;
        uint32_t* out;
        uint32_t row_shift;                 // KIND_ROW_T: log2 of the input row stride
        uint32_t col0;                      // KIND_COL: first column index j' of the tile
        uint32_t out0;                      // in-transform index of output (c = 0, k = 0): KIND_ROW_T k1_0 + mid * M_1, else 0
        uint32_t valid_c;                   // KIND_ROW_N: rows of this tile that exist
        uint32_t bidx;                      // KIND_ROW_T: the transform (row of the batch) this tile belongs to
    };

    // XCD-aware order: workgroups b and b+8 run on the same XCD (round-robin dispatch), and tiles 2x, 2x+1 share
    // 128-byte lines (16 columns x 4 B = half a line), so within every 16 consecutive virtual indices the pair
    // (2x, 2x+1) goes to workgroups x and x+8.  A bijection; placement only ever affects speed.
    static TOYNI_HD uint32_t tile_order(uint32_t v, uint32_t ntiles) {
        if ((ntiles & 15u) != 0) return v;
        const uint32_t s = v & 15u;
        return (v & ~15u) | ((s & 7u) << 1) | (s >> 3);
    }

    static TOYNI_HD Tile tile_of(const PassArgs& a, uint32_t bid) {
        Tile t;
        t.col0 = 0;
        t.out0 = 0;
        t.valid_c = C;
        t.row_shift = 0;
        t.bidx = 0;
        if (KIND == KIND_COL) {
            const uint32_t tiles_log = a.log_S - LC;
            const uint64_t prefix = (uint64_t)bid >> tiles_log;
            t.col0 = (bid & ((1u << tiles_log) - 1)) << LC;
            const uint64_t base = (prefix << (a.log_S + LM)) + t.col0;
            t.in = a.in + ((prefix << a.in_prefix_log) + t.col0);
            t.out = a.out + base;
            t.col0 += a.col_base;  // from here on col0 only feeds twiddle exponents
        } else if (KIND == KIND_ROW_T) {
            // interleaved: a tile's C virtual rows are C >> LQ consecutive k_1 times the Q transforms; b counts Q-groups
            const uint32_t mid = bid & ((1u << a.log_mid) - 1);
            const uint32_t k1_tiles_log = a.log_M1 - (LC - LQ_);
            const uint32_t k1_0 = ((bid >> a.log_mid) & ((1u << k1_tiles_log) - 1)) << (LC - LQ_);
            const uint64_t b = (uint64_t)bid >> (a.log_mid + k1_tiles_log);
            t.row_shift = a.log_n - a.log_M1;
            t.in = a.in + (((b << a.log_n) + ((uint64_t)k1_0 << t.row_shift) + ((uint64_t)mid << LM)) << LQ_);
            t.out = a.out + (((b << (SLAB_ ? a.out_prefix_log : a.log_n)) + k1_0 + ((uint64_t)mid << a.log_M1)) << LQ_);
            t.out0 = k1_0 + (mid << a.log_M1);
            if (SLAB_) t.bidx = (uint32_t)b;
        } else {
            const uint64_t row0 = (uint64_t)bid << LC;   // virtual rows (interleaved: Q per batch entry; C is a multiple of Q)
            t.in = a.in + (row0 << LM);
            t.out = a.out + (row0 << LM);
            const uint64_t left = a.rows_total - row0;
            t.valid_c = left < C ? (uint32_t)left : C;
        }
        return t;
    }

    // element offset of tile element (c, r) in the input (linear in c and r)
    // (interleaved row kinds: linear in r / k only, which is all the callers use -- c = (row << LQ) | q)
    static TOYNI_HD uint32_t in_offset(const PassArgs& a, const Tile& t, uint32_t c, uint32_t r) {
        if (KIND == KIND_COL) return (r << a.log_S) + c;
        if (KIND == KIND_ROW_T) return ((((c >> LQ_) << t.row_shift) + r) << LQ_) + (c & (Q - 1u));
        return ((((c >> LQ_) << LM) + r) << LQ_) + (c & (Q - 1u));
    }
    // element offset of finished element (c, natural sub-index k) in the output (linear in c and k)
    static TOYNI_HD uint32_t out_offset(const PassArgs& a, uint32_t c, uint32_t k) {
        if (KIND == KIND_COL) return (k << a.log_S) + c;
        if (KIND == KIND_ROW_T) return c + (k << (a.log_n - LM + LQ_));
        return ((((c >> LQ_) << LM) + k) << LQ_) + (c & (Q - 1u));
    }

    // ---- thread coordinates ----------------------------------------------------------------------------
    // step 1: KIND_COL lanes run over columns first (64-byte row segments), KIND_ROW_* over the contiguous row
    // (interleaved row kinds: the Q transforms of one position are the fastest lanes -- consecutive words in memory)
    static TOYNI_HD void coords1(uint32_t tid, uint32_t& c, uint32_t& lo) {
        if (KIND == KIND_COL) { c = tid & (C - 1); lo = tid >> LC; }
        else if (LQ_ == 0) { lo = tid & (E2 - 1); c = tid >> LE2; }
        else { lo = (tid >> LQ_) & (E2 - 1); c = ((tid >> (LQ_ + LE2)) << LQ_) | (tid & (Q - 1u)); }
    }
    // step 2, group g of the thread: lanes run over what is contiguous in the OUTPUT
    static TOYNI_HD void coords2(uint32_t tid, uint32_t g, uint32_t& c, uint32_t& hi) {
        const uint32_t gamma = tid + g * T;
        if (KIND == KIND_ROW_N) {
            if (LQ_ == 0) { hi = gamma & (E1 - 1); c = gamma >> LE1; }
            else {
                // A wave holds Q x 16 (q, hi) pairs.  Its store covers the output sub-indices bitrev(hi): with E1 = 32 the 16 consecutive
                // hi of a wave would be every OTHER output element (16-byte chunks at a 32-byte stride, the rest of each line written by
                // another wave at another time: measured 2.8 TB/s at n = 2^9).  Rotating the index left by one bit gives a wave the 16 hi
                // of one parity = 16 CONSECUTIVE outputs x Q words = 256 contiguous bytes (at the price of a 2-way bank conflict on the
                // step's LDS reads: the 16 even hi land on the even banks).
                uint32_t h = (gamma >> LQ_) & (E1 - 1);
                if (LE1 == 5) h = ((h << 1) | (h >> (LE1 - 1))) & (E1 - 1);
                hi = h;
                c = ((gamma >> (LQ_ + LE1)) << LQ_) | (gamma & (Q - 1u));
            }
        }
        else { c = gamma & (C - 1); hi = gamma >> LC; }
    }

    // KIND_COL inter-pass twiddle of a step-2 group: w_L^(j' k) with k = (b << LSH) | khi is A * G^b,
    // A = w_L^(j' khi) (times n^-1 on the first pass of an inverse), G = w_L^(j' << LSH): two table lookups per
    // group and a running product instead of a gather per element.  The lookups are split into an issue half (the
    // four table words) and a finish half, so the persistent kernel can issue them a tile ahead.
    struct Twiddle { uint32_t a0, g; };
    struct TwiddleRaw { uint32_t a_lo, a_hi, g_lo, g_hi; };
    template <int LSH>
    static TOYNI_HD TwiddleRaw group_twiddle_issue(const PassArgs& a, const Tile& t, uint32_t c, uint32_t khi) {
        TwiddleRaw r{0u, 0u, 0u, 0u};
        if (KIND == KIND_COL) {
            const uint32_t jcol = (t.col0 + c) >> LQ_;          // interleaved: Q consecutive columns share a twiddle column
            const uint32_t mask = (1u << a.tw_lowbits) - 1u;
            const uint32_t ea = jcol * khi, eg = jcol << LSH;   // < L <= 2^27
            r.a_lo = a.tw_lo[ea & mask];
            r.a_hi = a.tw_hi[ea >> a.tw_lowbits];
            r.g_lo = a.tw_lo[eg & mask];
            r.g_hi = a.tw_hi[eg >> a.tw_lowbits];
        } else if (a.cs_mode == 2u) {
            // output coset factor s^k of the group's first element (b = 0): the in-transform output index
            const uint32_t e0 = KIND == KIND_ROW_T ? t.out0 + (c >> LQ_) + (khi << (a.log_n - LM)) : khi;
            r.a_lo = a.cs_lo[e0 & ((1u << a.cs_lowbits) - 1u)];
            r.a_hi = a.cs_hi[e0 >> a.cs_lowbits];
        } else if (SLAB_ && KIND == KIND_ROW_T && a.cs_mode == 4u) {
            // slab form, inverse: output kk of row k1 = row0 + (transform index) times w_N^-(k1 kk); consecutive registers of a group
            // are (1 << LSH) << (log_n - LM) outputs apart, so the running factor is w_N^-(k1 * that) -- per tile, looked up like the seed
            const uint32_t k1 = a.row0 + t.bidx;
            const uint32_t mask = (1u << a.cs_lowbits) - 1u;
            const uint32_t e0 = (t.out0 + (c >> LQ_) + (khi << (a.log_n - LM))) * k1;   // < N <= 2^27
            const uint32_t eg = (k1 << LSH) << (a.log_n - LM);
            r.a_lo = a.cs_lo[e0 & mask];
            r.a_hi = a.cs_hi[e0 >> a.cs_lowbits];
            r.g_lo = a.cs_lo[eg & mask];
            r.g_hi = a.cs_hi[eg >> a.cs_lowbits];
        }
        return r;
    }
    static TOYNI_HD Twiddle group_twiddle_finish(const PassArgs& a, const TwiddleRaw& r) {
        Twiddle tw{0u, 0u};
        if (KIND == KIND_COL) {
            tw.a0 = mont_mul(r.a_hi, r.a_lo);
            if (a.scale) tw.a0 = mont_mul(tw.a0, a.scale);
            tw.g = mont_mul(r.g_hi, r.g_lo);
        } else if (a.cs_mode == 2u) {
            tw.a0 = mont_mul(r.a_hi, r.a_lo);
            if (KIND == KIND_ROW_N && a.scale) tw.a0 = mont_mul(tw.a0, a.scale);  // 1-pass inverse: n^-1 rides along
            tw.g = a.cs_g;
        } else if (SLAB_ && KIND == KIND_ROW_T && a.cs_mode == 4u) {
            tw.a0 = mont_mul(r.a_hi, r.a_lo);
            tw.g = mont_mul(r.g_hi, r.g_lo);
        }
        return tw;
    }
    template <int LSH>
    static TOYNI_HD Twiddle group_twiddle(const PassArgs& a, const Tile& t, uint32_t c, uint32_t khi) {
        return group_twiddle_finish(a, group_twiddle_issue<LSH>(a, t, c, khi));
    }

    // Finish and store the NB = 2^LB elements held for one (c, khi): register i holds natural sub-index
    // k = (bitrev(i) << LSH) | khi.
    template <int LB, int LSH>
    static TOYNI_HD void finish(const PassArgs& a, const Tile& t, uint32_t c, uint32_t khi, Twiddle twd, uint32_t (&x)[1 << LB]) {
        constexpr uint32_t NB = 1u << LB;
        // out_offset is linear in k: element b sits at off0 + b * step (bytes)
        const uint32_t off0 = out_offset(a, c, khi) << 2;
        const uint32_t step = (out_offset(a, 0u, 1u << LSH) - out_offset(a, 0u, 0u)) << 2;
        char* base = reinterpret_cast<char*>(t.out);
#if TOYNI_BUF
        // single-pass kinds keep the pointer form (measured: n = 2^10 loses 17 % with buffer stores, n = 2^4 / 2^5 half their rate with
        // buffer loads -- profiles/r02_ab_buffer.txt)
        const BufRsrc ws = buf_rsrc(base);
#define TOYNI_STORE(B, V) do { if constexpr (KIND != KIND_ROW_N) stb32<NT_>(ws, off0, (B) * step, (V)); \
                               else st32<NT_>(reinterpret_cast<uint32_t*>(base + (uint64_t)(B) * step), off0, (V)); } while (0)
#else
#define TOYNI_STORE(B, V) st32<NT_>(reinterpret_cast<uint32_t*>(base + (uint64_t)(B) * step), off0, (V))
#endif
        if (KIND == KIND_COL) {
            uint32_t tw = twd.a0;
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                TOYNI_STORE(b, mont_mul(x[cx_bitrev(b, LB)], tw));
                if (b + 1 < NB) { tw = mont_mul_lazy(tw, twd.g); TOYNI_PIN(tw); }
            }
        } else if (SLAB_ && KIND == KIND_ROW_T) {
            // slab form, inverse (toyni_ntt_slab_rows_device): the row's outputs go straight into the pieces layout -- sub-index k belongs
            // to piece k >> out_split_shift, which for register b is a compile-time constant shifted by a uniform amount: the piece
            // offset joins the store's scalar offset -- and are multiplied by w_N^-(k1 kk) (cs_mode 4) on the way
            uint32_t tw = twd.a0;
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                const uint32_t piece = ((b << LSH) >> a.out_split_shift) * (a.out_split_extra << 2);   // store b carries sub-index (b << LSH) | khi
                uint32_t v = x[cx_bitrev(b, LB)];
                if (a.cs_mode == 4u) {
                    v = mont_mul(v, tw);
                    if (b + 1 < NB) { tw = mont_mul_lazy(tw, twd.g); TOYNI_PIN(tw); }
                }
#if TOYNI_BUF
                stb32<NT_>(ws, off0, b * step + piece, v);
#else
                st32<NT_>(reinterpret_cast<uint32_t*>(base + (uint64_t)b * step + piece), off0, v);
#endif
            }
        } else if (a.cs_mode == 2u) {  // inverse coset transform: * s^k, k = k0 + b * (register step), running product
            uint32_t tw = twd.a0;
#if defined(__HIP_DEVICE_COMPILE__)
            // one thread per row: 16-byte stores, as in the plain case below (with 4-byte stores: 196 against 434 Gel/s at n = 32)
            if constexpr (!TWO_STEP && KIND == KIND_ROW_N && !NT_ && NB >= 4 && LQ_ == 0) {
#pragma unroll
                for (uint32_t q = 0; q < NB / 4; ++q) {
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        v[j] = mont_mul(x[cx_bitrev(4 * q + j, LB)], tw);
                        if (4 * q + j + 1 < NB) { tw = mont_mul_lazy(tw, twd.g); TOYNI_PIN(tw); }
                    }
                    u32x4_a4 w;
                    w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
                    *reinterpret_cast<u32x4_a4*>(base + off0 + 16u * q) = w;
                }
                return;
            }
#endif
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                TOYNI_STORE(b, mont_mul(x[cx_bitrev(b, LB)], tw));
                if (b + 1 < NB) { tw = mont_mul_lazy(tw, twd.g); TOYNI_PIN(tw); }
            }
        } else {
            const bool scaled = KIND == KIND_ROW_N && a.scale != 0u;  // 1-pass inverse only (multi-pass: the first pass scales)
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (!TWO_STEP && KIND == KIND_ROW_N && !NT_ && NB >= 4 && LQ_ == 0) {   // the thread's whole row, consecutive words: 16-byte stores
#pragma unroll
                for (uint32_t q = 0; q < NB / 4; ++q) {
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        v[j] = x[cx_bitrev(4 * q + j, LB)];
                        if (scaled) v[j] = mont_mul(v[j], a.scale);
                    }
                    u32x4_a4 w;
                    w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
                    *reinterpret_cast<u32x4_a4*>(base + off0 + 16u * q) = w;
                }
                return;
            }
#endif
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                uint32_t v = x[cx_bitrev(b, LB)];
                if (scaled) v = mont_mul(v, a.scale);
                TOYNI_STORE(b, v);
            }
        }
#undef TOYNI_STORE
    }

    // LE bits of radix-2 DIF butterflies on x[0..2^LE), register index bit s <-> sub-transform bit s + SHIFT.
    // Twiddle of the butterfly (i, i + 2^s): w_{2^(t+1)}^(low + q 2^SHIFT), t = s + SHIFT, q = i mod 2^s.
    //   SHIFT > 0: per-thread (`low` = the thread's bits below SHIFT), read from `tw1` = the packed stage table from
    //              stage SHIFT on (the persistent kernel keeps that slice in LDS: reads count on lgkmcnt, so they
    //              never wait behind the previous tile's HBM stores the way a vmcnt load would);
    //   SHIFT = 0: the same for every thread -> `uni` holds the top stage's 2^(LE-1) twiddles in SGPRs (the lower
    //              stages reuse them at a stride), and the q = 0 butterflies have twiddle 1 (plain subtract).
    template <int LE, int SHIFT, int S>
    static TOYNI_HD void stage_twiddles(uint32_t (&w)[1 << (LE - 1)], uint32_t (&nw)[1 << (LE - 1)], const uint32_t* tw1, uint32_t low,
                                        const uint32_t* uni) {
        constexpr int t = S + SHIFT;  // butterfly spans 2^(t+1) elements
#pragma unroll
        for (int q = 0; q < (1 << S); ++q) {
            // a lower stage's twiddles are a stride of the top stage's: w_{2^(t+1)}^q = w_{2^LE}^(q << (LE-1-t))
            w[q] = SHIFT == 0 ? uni[(uint32_t)q << (LE - 1 - S)] : tw1[(1u << t) - (1u << SHIFT) + low + ((uint32_t)q << SHIFT)];
            nw[q] = BB_P - w[q];
        }
    }
    // ---- radix-4 form of the wave-uniform stages (SHIFT = 0) ----
    // Two consecutive stages s, s-1 on the four elements a = x[i], b = x[i+d'], c = x[i+d], e = x[i+d+d'] (d = 2^s, d' = 2^(s-1),
    // q = i mod d'), with W = w_{2^(s+1)}:   s0 = a+c, s1 = b+e, d0 = a-c, d1 = b-e,
    //   x[i]      = s0 + s1                              x[i+d']   = (s0 - s1) W^(2q)
    //   x[i+d]    = d0 W^q + d1 W^(q+d')                  x[i+d+d'] = d0 W^(3q) - d1 W^(3q+d')
    // The two mixed outputs are ONE two-term Montgomery dot product each, so the group costs 5 canonical add/subs and 3
    // reductions = 33 instructions against 36 for its four radix-2 butterflies (119 against 135 issue cycles).  Every twiddle is a
    // power of the top-stage root, i.e. +-uni[k] with k and the sign known at compile time (W^(2^s) = -1): no extra table.  The
    // q = 0 groups keep the radix-2 form (27 instructions: three of their four butterflies have twiddle 1).
    // uni[k] = w_{2^LE}^k, k < 2^(LE-1).  U<LE>(uni, k, neg) = +-w_{2^LE}^k for k < 2^LE as a canonical Montgomery multiplier.
    template <int LE>
    static TOYNI_HD uint32_t uni_pow(const uint32_t* uni, uint32_t k, bool negate) {
        constexpr uint32_t HALF = 1u << (LE - 1);
        const bool neg = (k >= HALF) != negate;
        const uint32_t w = uni[k & (HALF - 1u)];
        return neg ? BB_P - w : w;
    }
    template <int LE, int S>
    static TOYNI_HD void stages_uniform_r4(uint32_t (&x)[1 << LE], const uint32_t* uni) {
        if constexpr (S >= 1) {
            constexpr uint32_t D = 1u << S, DP = 1u << (S - 1), G = 1u << (LE - 1 - S), QUARTER = 1u << (LE - 2);
#pragma unroll
            for (uint32_t i = 0; i < (1u << LE); ++i) {
                if (i & (D | DP)) continue;
                const uint32_t q = i & (DP - 1u);
                const uint32_t a = x[i], b = x[i + DP], c = x[i + D], e = x[i + D + DP];
                const uint32_t s0 = bb_add(a, c), s1 = bb_add(b, e), d0 = bb_sub(a, c);
                x[i] = bb_add(s0, s1);
                if (q == 0) {   // W^0 = 1: x[i+d'] = s0 - s1, and the mixed outputs are d0 +- (b - e) w_4 (27 instructions, as radix 2)
                    x[i + DP] = bb_sub(s0, s1);
                    const uint32_t t = mont_dot_sub(b, e, uni_pow<LE>(uni, QUARTER, false), uni_pow<LE>(uni, QUARTER, true));
                    x[i + D] = bb_add(d0, t);
                    x[i + D + DP] = bb_sub(d0, t);
                } else {
                    const uint32_t d1 = bb_sub(b, e);
                    x[i + DP] = mont_dot_sub(s0, s1, uni_pow<LE>(uni, 2u * q * G, false), uni_pow<LE>(uni, 2u * q * G, true));
                    x[i + D] = mont_dot2(d0, uni_pow<LE>(uni, q * G, false), d1, uni_pow<LE>(uni, q * G + QUARTER, false));
                    x[i + D + DP] = mont_dot2(d0, uni_pow<LE>(uni, 3u * q * G, false), d1, uni_pow<LE>(uni, 3u * q * G + QUARTER, true));
                }
            }
            stages_uniform_r4<LE, S - 2>(x, uni);
        } else if constexpr (S == 0) {   // one stage left: span 2, twiddle 1
#pragma unroll
            for (uint32_t i = 0; i < (1u << LE); i += 2) {
                const uint32_t u = x[i], v = x[i + 1];
                x[i] = bb_add(u, v);
                x[i + 1] = bb_sub(u, v);
            }
        }
    }

    // ---- radix-4 form of the per-thread stages (SHIFT > 0): the same group, twiddles from tables ----
    // With e = low + (q << SHIFT) the five multipliers of a group are W^e, W^(e+2^(t-1)) (stage t = s + SHIFT), V^e = W^(2e) (stage
    // t-1), W^(3e) and -W^(3e+2^(t-1)).  The first three are entries of the packed stage table; the last two come from its radix-4
    // companion `tw3` (same size, PassArgs::stage_tw3) -- W^(3e) wraps past W^(2^t) = -1 at a thread-dependent point, so deriving
    // it from the stage table would cost a sign fix-up per thread and twiddle.  Besides the 3 instructions per group this form needs
    // ONE negated twiddle per q (for the dot_sub) where two radix-2 stages need three.
    // tw1: the stage table from stage SHIFT on; tw3: the companion from block SHIFT + 1 on.
    template <int LE, int SHIFT, int S>
    static TOYNI_HD void stages_thread_r4(uint32_t (&x)[1 << LE], const uint32_t* tw1, const uint32_t* tw3, uint32_t low) {
        // `low` is made opaque: knowing low < 2^SHIFT the compiler turns every table address into (low << 2) | constant -- one
        // VGPR per table entry, all loop-invariant, all hoisted out of the tile loop (51 of them: spills).  As an opaque value the
        // addresses stay (one register) + (immediate offset).
        TOYNI_PIN(low);
        if constexpr (S >= 1) {
            constexpr int t = S + SHIFT;
            constexpr uint32_t D = 1u << S, DP = 1u << (S - 1), HALF = 1u << (t - 1);
            const uint32_t* t1 = tw1 + ((1u << t) - (1u << SHIFT));
            const uint32_t* t2 = tw1 + (HALF - (1u << SHIFT));
            const uint32_t* t3 = tw3 + ((1u << t) - (2u << SHIFT));
            // q outermost: the five multipliers of one q are live only while its 2^(LE-1-S) groups are processed (loading all
            // DP sets up front, as the radix-2 form does with its 2 x 16, would put 6 x DP + 32 values in flight: spills at DP = 8)
            // software-pipelined by one q: the next q's five table reads are issued before this q's groups are computed, and a
            // scheduling fence per q keeps the compiler from hoisting all of them to the top
            uint32_t n1 = t1[low], n1p = t1[low + HALF], n2 = t2[low], n3 = t3[low], n3n = t3[HALF + low];
#pragma unroll
            for (uint32_t q = 0; q < DP; ++q) {
                const uint32_t w1 = n1, w1p = n1p, w2 = n2, nw2 = BB_P - n2, w3 = n3, w3n = n3n;
                if (q + 1 < DP) {
                    const uint32_t e1 = low + ((q + 1) << SHIFT);
                    n1 = t1[e1]; n1p = t1[e1 + HALF]; n2 = t2[e1]; n3 = t3[e1]; n3n = t3[HALF + e1];
                }
                TOYNI_SCHED_FENCE();
#pragma unroll
                for (uint32_t hi = 0; hi < (1u << (LE - 1 - S)); ++hi) {
                    const uint32_t i = (hi << (S + 1)) | q;
                    const uint32_t a = x[i], b = x[i + DP], c = x[i + D], e = x[i + D + DP];
                    const uint32_t s0 = bb_add(a, c), s1 = bb_add(b, e), d0 = bb_sub(a, c), d1 = bb_sub(b, e);
                    x[i] = bb_add(s0, s1);
                    x[i + DP] = mont_dot_sub(s0, s1, w2, nw2);
                    x[i + D] = mont_dot2(d0, w1, d1, w1p);
                    x[i + D + DP] = mont_dot2(d0, w3, d1, w3n);
                }
            }
            stages_thread_r4<LE, SHIFT, S - 2>(x, tw1, tw3, low);
        } else if constexpr (S == 0) {   // one stage left: span 2^(SHIFT+1), twiddle w_{2^(SHIFT+1)}^low
            const uint32_t w = tw1[low], nw = BB_P - w;
#pragma unroll
            for (uint32_t i = 0; i < (1u << LE); i += 2) {
                const uint32_t u = x[i], v = x[i + 1];
                x[i] = bb_add(u, v);
                x[i + 1] = mont_dot_sub(u, v, w, nw);
            }
        }
    }

    // R4T: the caller supplies the radix-4 companion table `tw3` for the per-thread stages (SHIFT > 0)
    template <int LE, int SHIFT, bool R4T = false>
    static TOYNI_HD void stages(uint32_t (&x)[1 << LE], const uint32_t* tw1, uint32_t low, const uint32_t* uni, const uint32_t* tw3 = nullptr) {
#if !defined(TOYNI_NO_RADIX4)
        if constexpr (R4T && SHIFT > 0 && LE >= 2) {
            stages_thread_r4<LE, SHIFT, LE - 1>(x, tw1, tw3, low);
            return;
        }
        if constexpr (SHIFT == 0 && LE >= 2) {
            stages_uniform_r4<LE, LE - 1>(x, uni);
            return;
        }
#endif
#pragma unroll
        for (int s = LE - 1; s >= 0; --s) {
            const int d = 1 << s;
            const int t = s + SHIFT;  // butterfly spans 2^(t+1) elements
            uint32_t w[1 << (LE - 1)], nw[1 << (LE - 1)];
#pragma unroll
            for (int q = 0; q < d; ++q) {
                // a lower stage's twiddles are a stride of the top stage's: w_{2^(t+1)}^q = w_{2^LE}^(q << (LE-1-t))
                w[q] = SHIFT == 0 ? uni[(uint32_t)q << (LE - 1 - s)] : tw1[(1u << t) - (1u << SHIFT) + low + ((uint32_t)q << SHIFT)];
                nw[q] = BB_P - w[q];
            }
#pragma unroll
            for (int i = 0; i < (1 << LE); ++i) {
                if (i & d) continue;
                const uint32_t u = x[i], v = x[i + d];
                x[i] = bb_add(u, v);
                if (SHIFT == 0 && (i & (d - 1)) == 0) x[i + d] = bb_sub(u, v);
                else x[i + d] = mont_dot_sub(u, v, w[i & (d - 1)], nw[i & (d - 1)]);
            }
        }
    }

    // The same for the first pass of a low-degree extension (KIND_COL step 1 only): on entry only registers
    // i < 2^(LE-LZ) can be nonzero, so in the top LZ stages every butterfly has a zero partner, (u, 0) -> (u, u * w):
    // one multiply, no add, and the pairs whose inputs are both still zero are skipped.  Stage S as a template
    // parameter (compile-time recursion), so every register index is a constant whatever the unroller decides.
    template <int LE, int SHIFT, int LZ, int S>
    static TOYNI_HD void stages_lz(uint32_t (&x)[1 << LE], const uint32_t* tw1, uint32_t low, const uint32_t* uni) {
        constexpr int LZS = LZ < LE ? LZ : LE;
        constexpr int d = 1 << S;
        uint32_t w[1 << (LE - 1)], nw[1 << (LE - 1)];
        stage_twiddles<LE, SHIFT, S>(w, nw, tw1, low, uni);
        if constexpr (S >= LE - LZS) {
            // live inputs of this stage: bits [LE-LZS, S) of the register index are zero (the bits above S were filled
            // in by the earlier degenerate stages)
#pragma unroll
            for (int i = 0; i < (1 << LE); ++i) {
                if ((i & d) || ((i >> (LE - LZS)) & ((1 << (S - (LE - LZS))) - 1))) continue;
                x[i + d] = (SHIFT == 0 && (i & (d - 1)) == 0) ? x[i] : mont_mul(x[i], w[i & (d - 1)]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < (1 << LE); ++i) {
                if (i & d) continue;
                const uint32_t u = x[i], v = x[i + d];
                x[i] = bb_add(u, v);
                if (SHIFT == 0 && (i & (d - 1)) == 0) x[i + d] = bb_sub(u, v);
                else x[i + d] = mont_dot_sub(u, v, w[i & (d - 1)], nw[i & (d - 1)]);
            }
        }
        if constexpr (S > 0) stages_lz<LE, SHIFT, LZ, S - 1>(x, tw1, low, uni);
    }

    // step-1 slice of the packed stage table: stages LE2 .. LM-1 = words [2^LE2 - 1, 2^LM - 1)
    static constexpr uint32_t TW1_WORDS = TWO_STEP ? M - E2 : 0;
    static TOYNI_HD const uint32_t* tw1_global(const PassArgs& a) { return a.stage_tw + (E2 - 1u); }
    // step-1 slice of the radix-4 companion table: blocks LE2 + 1 .. LM - 1 = words [2^(LE2+1) - 2, 2^LM - 2)
#if defined(TOYNI_NO_RADIX4) || defined(TOYNI_NO_RADIX4_THREAD)   // A/B builds
    static constexpr bool STEP1_R4 = false;
#else
    static constexpr bool STEP1_R4 = TWO_STEP && LE1 >= 2 && LC >= 4;   // the 8-wide shapes keep radix 2: the second table slice would cost
#endif                                                                  // them their fourth workgroup per CU (LDS)
    static constexpr uint32_t TW3_WORDS = STEP1_R4 ? M - 2u * E2 : 0;
    static TOYNI_HD const uint32_t* tw3_global(const PassArgs& a) { return a.stage_tw3 + (2u * E2 - 2u); }

    // the wave-uniform twiddles of the last LU stage bits: w_{2^LU}^q, q < 2^(LU-1) = stage LU-1 of the packed table
    static constexpr int LU = TWO_STEP ? LE2 : LE1;
    static constexpr uint32_t NU = 1u << (LU - 1);
    struct Uniform { uint32_t w[NU]; };
    static TOYNI_HD Uniform load_uniform(const PassArgs& a) {
        Uniform u;
#pragma unroll
        for (uint32_t q = 0; q < NU; ++q) u.w[q] = TOYNI_UNIFORM(a.stage_tw[NU - 1u + q]);
        return u;
    }

    // ---- the pieces of a pass, in the order a workgroup runs them ---------------------------------------
    // (a) HBM -> registers: E1 elements of the thread's column / row, r = lo + i * E2
    //     (registers [I0, I1) only: the persistent kernel prefetches a leading part of the next tile)
    //     LZ > 0: only rows r < M >> LZ exist in the input (the rest is the implied zero padding): registers
    //     i >= E1 >> LZ are not loaded, and when LZ >= LE1 the one remaining load is guarded by lo < nz_rows.
    template <uint32_t I0 = 0, uint32_t I1 = E1, int LZ = 0>
    static TOYNI_HD void load_tile(const PassArgs& a, const Tile& t, uint32_t tid, uint32_t (&x)[E1]) {
        static_assert(LZ == 0 || KIND == KIND_COL, "zero-padded input is a first (column) pass feature");
        constexpr uint32_t NZ = E1 >> (LZ < LE1 ? LZ : LE1);
        uint32_t c, lo;
        coords1(tid, c, lo);
        const bool live = (KIND != KIND_ROW_N || c < t.valid_c)  // only single-pass row tiles can be ragged
                          && (LZ < LE1 || lo < a.nz_rows);
        // in_offset is linear in r: register i sits at off0 + i * step (bytes)
        const uint32_t off0 = in_offset(a, t, c, lo) << 2;
        const uint32_t step = (in_offset(a, t, 0u, E2) - in_offset(a, t, 0u, 0u)) << 2;
        // register i: (uniform base + i * uniform step) + one per-thread offset -> SGPR pointer math, a single VGPR
        const char* base = reinterpret_cast<const char*>(t.in);
        if (SLAB_ && KIND == KIND_COL && LZ == 0) {
            // slab form, forward (toyni_ntt_slab_rows_device): the tile's rows live in the pieces layout -- row r belongs to piece
            // r >> in_split_shift (>= LE2 bits, so the piece of register i does not depend on the thread): a uniform term per register
#pragma unroll
            for (uint32_t i = I0; i < I1; ++i) {
                const uint32_t piece = ((i << LE2) >> a.in_split_shift) * (a.in_split_extra << 2);
                x[i] = ld32<NT_>(reinterpret_cast<const uint32_t*>(base + (uint64_t)i * step + piece), off0);
            }
            return;
        }
        if (live) {
#if defined(__HIP_DEVICE_COMPILE__)
            // one thread per row (single-step shapes): its E1 words are consecutive -- four per load instead of one (a wave's scalar
            // loads each touch 32 different lines; measured at 2^28 elements: n = 2^4 2.9 -> see profiles/r02_sweep.txt)
            if constexpr (!TWO_STEP && KIND != KIND_COL && !NT_ && LZ == 0 && I0 == 0 && I1 == E1 && E1 >= 4 && LQ_ == 0) {
#pragma unroll
                for (uint32_t q = 0; q < E1 / 4; ++q) {
                    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(base + off0 + 16u * q);
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
                return;
            }
#endif
#if TOYNI_BUF
            if constexpr (KIND != KIND_ROW_N && !(LE1 == 5 && LE2 == 5)) {   // measured: the 32 x 32 shapes lose 5.5 % with buffer loads
                const BufRsrc rs = buf_rsrc(base);
#pragma unroll
                for (uint32_t i = I0; i < I1; ++i) x[i] = i < NZ ? ldb32<NT_>(rs, off0, i * step) : 0u;
            } else
#endif
            {
#pragma unroll
                for (uint32_t i = I0; i < I1; ++i) x[i] = i < NZ ? ld32<NT_>(reinterpret_cast<const uint32_t*>(base + (uint64_t)i * step), off0) : 0u;
            }
        } else {
#pragma unroll
            for (uint32_t i = I0; i < I1; ++i) x[i] = 0u;
        }
    }

    // (a') forward coset transform (cs_mode 1): the thread's inputs are x[j0 + i * dj]; scale them by s^(j0 + i dj) =
    //      A * G^i -- one table lookup pair per thread and a running product (G = a.cs_g is uniform)
    struct InSeedRaw { uint32_t lo, hi; };
    static TOYNI_HD InSeedRaw in_seed_issue(const PassArgs& a, const Tile& t, uint32_t tid) {
        InSeedRaw r{0u, 0u};
        if (KIND != KIND_ROW_T && a.cs_mode == 1u) {
            uint32_t c, lo;
            coords1(tid, c, lo);
            const uint32_t j0 = KIND == KIND_COL ? ((lo << a.log_S) + t.col0 + c) >> LQ_ : lo;  // in-transform index of register 0
            r.lo = a.cs_lo[j0 & ((1u << a.cs_lowbits) - 1u)];
            r.hi = a.cs_hi[j0 >> a.cs_lowbits];
        }
        return r;
    }
    template <int LZ = 0>
    static TOYNI_HD void in_scale(const PassArgs& a, const InSeedRaw& r, uint32_t (&x)[E1]) {
        constexpr uint32_t NZ = E1 >> (LZ < LE1 ? LZ : LE1);  // zero-padded input: only these registers hold data
        if (KIND != KIND_ROW_T && a.cs_mode == 1u) {
            uint32_t tw = mont_mul(r.hi, r.lo);
#pragma unroll
            for (uint32_t i = 0; i < NZ; ++i) {
                x[i] = mont_mul(x[i], tw);
                if (i + 1 < NZ) { tw = mont_mul_lazy(tw, a.cs_g); TOYNI_PIN(tw); }
            }
        }
    }

    // (b) the LE1 high-bit stages in registers, then park the tile in LDS (two-step) or finish (single-step)
    template <int LZ = 0>
    static TOYNI_HD void step1(const PassArgs& a, const Tile& t, uint32_t tid, uint32_t (&x)[E1], uint32_t* lds, const Uniform& uni,
                               const uint32_t* tw1, const uint32_t* tw3) {
        static_assert(LZ == 0 || TWO_STEP, "zero-padded input needs a two-step pass");
        uint32_t c, lo;
        coords1(tid, c, lo);
        if constexpr (LZ == 0) stages<LE1, LE2, STEP1_R4>(x, tw1, lo, uni.w, tw3);
        else stages_lz<LE1, LE2, LZ, LE1 - 1>(x, tw1, lo, uni.w);
        if (TWO_STEP) {
            const uint32_t base = lds_word(c, 0u, lo);
#pragma unroll
            for (uint32_t i = 0; i < E1; ++i) lds[base + i * HI_STRIDE] = x[i];
        } else {
            if (KIND != KIND_ROW_N || c < t.valid_c) finish<LE1, 0>(a, t, c, 0u, group_twiddle<0>(a, t, c, 0u), x);
        }
    }

    // (c) inter-pass twiddle seeds of the thread's step-2 groups: table lookups, issued early (see the kernel)
    struct SeedsRaw { TwiddleRaw g[G2 ? G2 : 1]; };
    struct Seeds { Twiddle g[G2 ? G2 : 1]; };
    static TOYNI_HD SeedsRaw seeds_issue(const PassArgs& a, const Tile& t, uint32_t tid) {
        SeedsRaw s;
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) {
            uint32_t c, hi;
            coords2(tid, g, c, hi);
            s.g[g] = group_twiddle_issue<LE1>(a, t, c, bitrev32(hi, LE1));
        }
        return s;
    }
    static TOYNI_HD Seeds seeds_finish(const PassArgs& a, const SeedsRaw& r) {
        Seeds s;
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) s.g[g] = group_twiddle_finish(a, r.g[g]);
        return s;
    }

    // (d) after the barrier: LDS -> registers, the LE2 low-bit stages (wave-uniform twiddles), twiddle, HBM store
    static TOYNI_HD void step2(const PassArgs& a, const Tile& t, uint32_t tid, const uint32_t* lds, const Seeds& seeds, const Uniform& uni) {
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) {
            uint32_t c, hi;
            coords2(tid, g, c, hi);
            uint32_t x[E2];
            const uint32_t base = lds_word(c, hi, 0u);
#pragma unroll
            for (uint32_t i = 0; i < E2; ++i) x[i] = lds[base + i * LOW_STRIDE];
            stages<LE2, 0>(x, nullptr, 0u, uni.w);
            if (KIND != KIND_ROW_N || c < t.valid_c) finish<LE2, LE1>(a, t, c, bitrev32(hi, LE1), seeds.g[g], x);
        }
    }

    // Whole tile, one thread, in program order (single-step kinds; tests/emu runs the two-step kinds as
    // phase1 for every thread, then phase2 for every thread -- the pass' one data barrier).
    template <int LZ = 0>
    static TOYNI_HD void phase1(const PassArgs& a, uint32_t tile_id, uint32_t tid, uint32_t* lds) {
        const Tile t = tile_of(a, tile_id);
        uint32_t x[E1];
        load_tile<0, E1, LZ>(a, t, tid, x);
        in_scale<LZ>(a, in_seed_issue(a, t, tid), x);
        step1<LZ>(a, t, tid, x, lds, load_uniform(a), tw1_global(a), tw3_global(a));
    }
    static TOYNI_HD void phase2(const PassArgs& a, uint32_t tile_id, uint32_t tid, const uint32_t* lds) {
        const Tile t = tile_of(a, tile_id);
        step2(a, t, tid, lds, seeds_finish(a, seeds_issue(a, t, tid)), load_uniform(a));
    }
};

// ---- three-step passes: the latency configuration (a single transform, or a handful) ----
// A lone n = 2^20 transform is 32 tiles of the 32-wide streaming shape and 128 of the 8-wide one: at most half the chip
// works, and every wave walks 32 elements through ~1 800 instructions on its own.  Pass3 cuts the same M-point
// sub-transform into THREE register steps of LE1 + LE2 + LE3 stage bits with only E = 2^LE1 <= 16 elements per thread and
// tiles of C = 4 columns / rows: 4x the workgroups (every CU gets one at n = 2^20) and half the serial work per wave.
// The data of such launches is cache-resident, so the narrow (16-byte) row segments cost little -- streaming launches keep
// the 32-wide two-step shapes above.
//   position r = (a << (LE2+LE3)) | (b << LE3) | d      a: LE1 bits (step 1, registers), b: LE2 bits (step 2), d: LE3 bits (step 3)
//   step 1: thread <-> (c, lo = (b, d)), registers over a   -- per-thread twiddles, LDS write
//   step 2: group  <-> (c, a, d),        registers over b   -- per-thread twiddles, in place in LDS
//   step 3: group  <-> (c, hm = (a, b)), registers over d   -- uniform twiddles (SGPRs), inter-pass twiddle / scaling, HBM store
//   output sub-index of (a, b, d): k = rev(d) << (LE1+LE2) | rev(b) << LE1 | rev(a) = rev(d) << (LE1+LE2) | rev_{LE1+LE2}(hm)
// LDS: KIND_COL word(c, r) = r C + c + (r >> LE3) PADC; KIND_ROW_T word(c, r) = c PITCH + r + (r >> LE3): every access of a
// thread is one base register + a compile-time immediate; steps 1 and 2 are conflict-free, step 3 of the column kind too.
// LQ_ = 2 (round 4): the interleaved (Ext, AoS) form, see Pass.  A 4-wide tile is then exactly ONE element's four coordinates: the
// column kind needs only the twiddle column (column >> LQ); the row kind runs its four rows as the four coordinates q of one k_1
// (row r of coordinate q at word 4 r + q: a wave's loads are 4-byte lanes at a 16-byte stride, the four waves of a row tile sharing
// their lines -- these launches are cache-resident by construction -- and its stores the same 16-byte chunks as the base form's).
//
// Round 5: the STREAMING three-step shape (LE1 = 5: 32 elements per thread, tiles of 16 rows / columns, 1024 threads): a 2048-point
// pass at the occupancy of the 32-wide 1024-point two-step shape (one workgroup per CU, ~150 KiB of LDS), run by the prefetching
// kernel ntt_pass3s_kernel.  With it n = 2^21 is TWO sweeps (1024-point column pass + 2048-point closing pass) instead of three
// 128-point ones.  Differences from the latency shapes: the step-1 stages take the radix-4 form (their table slices live in LDS),
// the step-2 twiddles of a thread are the same for every group and every tile and stay in registers (Tw2), stores take the
// buffer-resource form, and the row pitch is 2 mod 32 so that a step-3 group of 16 rows x 2 positions reads 32 different banks.
template <int KIND, int LE1, int LE2, int LE3, int LC, bool NT_ = false, int LQ_ = 0>
struct Pass3 {
    static_assert(KIND == KIND_COL || KIND == KIND_ROW_T, "passes of multi-pass plans");
    static_assert(LQ_ == 0 || (LQ_ == 2 && (LC == 2 || (LC == 4 && LE1 == 5))),
                  "interleaved tiles: one element's four coordinates (latency), or four rows / elements x four coordinates (streaming shapes)");
    static constexpr int LQ = LQ_;
    static_assert(LE1 >= LE2 && LE2 >= LE3 && LE3 >= 1 && LE1 <= 5, "step sizes");
    static constexpr int STEPS = 3;
    static constexpr bool STREAM = LE1 == 5;
    static constexpr int LM = LE1 + LE2 + LE3, LLO = LE2 + LE3, LHM = LE1 + LE2;
    static constexpr int IN_STEP_LOG = LLO, OUT_STEP_LOG = LHM;
    static constexpr bool NT = NT_;
    static constexpr uint32_t M = 1u << LM, E = 1u << LE1, E2 = 1u << LE2, E3 = 1u << LE3, C = 1u << LC;
    static constexpr uint32_t T = C << LLO;            // threads per workgroup = C * M / E
    static constexpr uint32_t G2 = E / E2, G3 = E / E3;  // step-2 / step-3 groups per thread
    static_assert(T >= 64 && T <= 1024, "workgroup size");
    using Stg = Pass<KIND_ROW_N, 5, 5, 3>;              // the stage code is shared (static members, independent of the shape)

    static constexpr uint32_t PADC = C < 32 ? C : 0;
    static constexpr uint32_t PITCH = STREAM ? (M + (M >> LE3)) + ((2u + 32u - ((M + (M >> LE3)) & 31u)) & 31u) : ((M + (M >> LE3)) | 1u);
    static constexpr uint32_t LDS_WORDS = KIND == KIND_COL ? M * C + (M >> LE3) * PADC : C * PITCH;
    static constexpr uint32_t TW_WORDS = M - E3;        // stages LE3 .. LM-1 of the packed stage table, kept in LDS (latency shapes)
    // streaming shapes: only step 1 reads its twiddles from LDS -- stages LLO .. LM-1 of the packed table and blocks LLO+1 .. LM-1 of
    // its radix-4 companion
    static constexpr bool STEP1_R4 = STREAM;
    static constexpr uint32_t TW1S_WORDS = STREAM ? M - (1u << (LE2 + LE3)) : 0u;
    static constexpr uint32_t TW3S_WORDS = STEP1_R4 ? M - (2u << (LE2 + LE3)) : 0u;
    static constexpr uint32_t MIN_WAVES = 4;
    // LDS word distance between two consecutive registers of a thread in step 1 / 2 / 3
    static constexpr uint32_t STRIDE1 = KIND == KIND_COL ? (C << LLO) + (PADC << LE2) : (1u << LLO) + (1u << LE2);
    static constexpr uint32_t STRIDE2 = KIND == KIND_COL ? (C << LE3) + PADC : E3 + 1u;
    static constexpr uint32_t STRIDE3 = KIND == KIND_COL ? C : 1u;
    // Interleaved streaming row shape (LQ = 2, 16 virtual rows c = (row << 2) | q): virtual row c is parked in LDS row 4 q + row.  With
    // PITCH = 2 mod 32 a step-1 group of 32 lanes (4 coordinates x 8 consecutive positions of one row) then writes banks
    // 8 q + 2 row + position -- 32 different ones -- and a step-3 group (the 16 virtual rows x 2 positions) reads all the even banks
    // for one position and all the odd ones for the other (positions are 9 words apart).
    static TOYNI_HD uint32_t lds_row(uint32_t c) {
        if (LQ_ == 0 || KIND == KIND_COL || C != 16u) return c;
        return ((c & 3u) << 2) + (c >> 2);
    }
    static TOYNI_HD uint32_t lds_word(uint32_t c, uint32_t r) {
        return KIND == KIND_COL ? r * C + c + (r >> LE3) * PADC : lds_row(c) * PITCH + r + (r >> LE3);
    }

    struct Tile {
        const uint32_t* in;
        uint32_t* out;
        uint32_t row_shift, col0, out0;
    };
    // XCD-aware order: G = 32 / C consecutive tiles share 128-byte lines; within every 8 G consecutive virtual indices they
    // go to workgroups p, p + 8, ... (one XCD under round-robin dispatch).  A bijection; placement only ever affects speed.
    static TOYNI_HD uint32_t tile_order(uint32_t v, uint32_t ntiles) {
        constexpr uint32_t G = C < 32 ? 32u / C : 1u;
        if (G == 1 || (ntiles & (8u * G - 1u)) != 0) return v;
        const uint32_t s = v & (8u * G - 1u);
        return (v & ~(8u * G - 1u)) | ((s & 7u) * G + (s >> 3));
    }
    static TOYNI_HD Tile tile_of(const PassArgs& a, uint32_t bid) {
        Tile t;
        t.col0 = 0;
        t.out0 = 0;
        t.row_shift = 0;
        if (KIND == KIND_COL) {
            const uint32_t tiles_log = a.log_S - LC;
            const uint64_t prefix = (uint64_t)bid >> tiles_log;
            t.col0 = (bid & ((1u << tiles_log) - 1)) << LC;
            t.in = a.in + ((prefix << a.in_prefix_log) + t.col0);
            t.out = a.out + ((prefix << (a.log_S + LM)) + t.col0);
            t.col0 += a.col_base;
        } else {
            const uint32_t mid = bid & ((1u << a.log_mid) - 1);
            const uint32_t k1_tiles_log = a.log_M1 - (LC - LQ_);
            const uint32_t k1_0 = ((bid >> a.log_mid) & ((1u << k1_tiles_log) - 1)) << (LC - LQ_);
            const uint64_t b = (uint64_t)bid >> (a.log_mid + k1_tiles_log);
            t.row_shift = a.log_n - a.log_M1;
            t.in = a.in + (((b << a.log_n) + ((uint64_t)k1_0 << t.row_shift) + ((uint64_t)mid << LM)) << LQ_);
            t.out = a.out + (((b << a.log_n) + k1_0 + ((uint64_t)mid << a.log_M1)) << LQ_);
            t.out0 = k1_0 + (mid << a.log_M1);
        }
        return t;
    }
    // (interleaved row kind: c = (row << LQ) | q; linear in r / k, which is all the callers use)
    static TOYNI_HD uint32_t in_offset(const PassArgs& a, const Tile& t, uint32_t c, uint32_t r) {
        return KIND == KIND_COL ? (r << a.log_S) + c : ((((c >> LQ_) << t.row_shift) + r) << LQ_) + (c & ((1u << LQ_) - 1u));
    }
    static TOYNI_HD uint32_t out_offset(const PassArgs& a, uint32_t c, uint32_t k) {
        return KIND == KIND_COL ? (k << a.log_S) + c : c + (k << (a.log_n - LM + LQ_));
    }

    // thread coordinates: lanes run over what is contiguous in HBM (columns / the row) in step 1 and over the LDS-contiguous
    // direction in step 2; in step 3 over the tile's columns / rows (contiguous in the output)
    static TOYNI_HD void coords1(uint32_t tid, uint32_t& c, uint32_t& lo) {
        if (KIND == KIND_COL) { c = tid & (C - 1); lo = tid >> LC; }
        else if (LQ_ == 0 || C == 4u) { lo = tid & ((1u << LLO) - 1u); c = tid >> LLO; }
        // interleaved rows of more than one element: the coordinates of one position are the fastest lanes (consecutive words in memory)
        else { lo = (tid >> LQ_) & ((1u << LLO) - 1u); c = ((tid >> (LQ_ + LLO)) << LQ_) | (tid & ((1u << LQ_) - 1u)); }
    }
    // Streaming row kind: step 1 parks row c = (wave index) -- coords1 puts one row on 2^LLO = 64 consecutive lanes -- and step 2 of a
    // row touches that row only, so each wave runs step 2 on ITS OWN row: the step-1 -> step-2 exchange is wave-local (a wave's LDS
    // operations execute in order) and the pass needs one data barrier per tile instead of two.
    // (not the interleaved form: there a wave holds 16 positions of each of the four coordinates' rows)
    static constexpr bool WAVE_LOCAL2 = STREAM && KIND == KIND_ROW_T && LE2 + LE3 == 6 && LQ_ == 0;
    static TOYNI_HD void coords2(uint32_t tid, uint32_t g, uint32_t& c, uint32_t& aa, uint32_t& d) {
        if (WAVE_LOCAL2) {
            const uint32_t l = tid & 63u;
            c = tid >> 6;
            d = l & (E3 - 1);
            aa = (l >> LE3) + g * (64u >> LE3);
            return;
        }
        const uint32_t gamma = tid + g * T;
        if (KIND == KIND_COL) { c = gamma & (C - 1); d = (gamma >> LC) & (E3 - 1); aa = gamma >> (LC + LE3); }
        else { d = gamma & (E3 - 1); aa = (gamma >> LE3) & (E - 1); c = gamma >> (LE3 + LE1); }
    }
    static TOYNI_HD void coords3(uint32_t tid, uint32_t g, uint32_t& c, uint32_t& hm) {
        const uint32_t gamma = tid + g * T;
        c = gamma & (C - 1);
        hm = gamma >> LC;
    }

    static constexpr uint32_t NU = E3 / 2 ? E3 / 2 : 1;
    struct Uniform { uint32_t w[NU]; };
    static TOYNI_HD Uniform load_uniform(const PassArgs& a) {
        Uniform u;
#pragma unroll
        for (uint32_t q = 0; q < NU; ++q) u.w[q] = TOYNI_UNIFORM(a.stage_tw[NU - 1u + q]);
        return u;
    }
    // twiddle slice kept in LDS: words [E3 - 1, M - 1) of the packed stage table; `tw` below points at its first word
    static TOYNI_HD const uint32_t* tw_global(const PassArgs& a) { return a.stage_tw + (E3 - 1u); }
    // streaming shapes: the step-1 slices (the stage table from stage LLO on; the companion from block LLO + 1 on)
    static TOYNI_HD const uint32_t* tw1s_global(const PassArgs& a) { return a.stage_tw + ((1u << LLO) - 1u); }
    static TOYNI_HD const uint32_t* tw3s_global(const PassArgs& a) { return a.stage_tw3 + ((2u << LLO) - 2u); }

    // ---- step 1: HBM -> registers (E elements: r = lo + (i << LLO)), coset input scaling, LE1 stages, park in LDS ----
    struct InSeedRaw { uint32_t lo, hi; };
    static TOYNI_HD InSeedRaw in_seed_issue(const PassArgs& a, const Tile& t, uint32_t tid) {
        InSeedRaw r{0u, 0u};
        if (KIND == KIND_COL && a.cs_mode == 1u) {
            uint32_t c, lo;
            coords1(tid, c, lo);
            const uint32_t j0 = ((lo << a.log_S) + t.col0 + c) >> LQ_;
            r.lo = a.cs_lo[j0 & ((1u << a.cs_lowbits) - 1u)];
            r.hi = a.cs_hi[j0 >> a.cs_lowbits];
        }
        return r;
    }
    template <int LZ = 0>
    static TOYNI_HD void load_tile(const PassArgs& a, const Tile& t, uint32_t tid, uint32_t (&x)[E]) {
        static_assert(LZ == 0 || KIND == KIND_COL, "zero-padded input is a first (column) pass feature");
        constexpr uint32_t NZ = E >> (LZ < LE1 ? LZ : LE1);
        uint32_t c, lo;
        coords1(tid, c, lo);
        const bool live = LZ < LE1 || lo < a.nz_rows;
        const uint32_t off0 = in_offset(a, t, c, lo) << 2;
        const uint32_t step = (in_offset(a, t, 0u, 1u << LLO) - in_offset(a, t, 0u, 0u)) << 2;
        const char* base = reinterpret_cast<const char*>(t.in);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i)
            x[i] = (live && i < NZ) ? ld32<NT_>(reinterpret_cast<const uint32_t*>(base + (uint64_t)i * step), off0) : 0u;
    }
    // tw1: the packed stage table from stage LLO on; tw3: its radix-4 companion from block LLO + 1 on (streaming shapes only)
    template <int LZ = 0>
    static TOYNI_HD void step1(const PassArgs& a, const InSeedRaw& seed, uint32_t tid, uint32_t (&x)[E], uint32_t* lds, const uint32_t* tw1,
                               const uint32_t* tw3 = nullptr) {
        constexpr uint32_t NZ = E >> (LZ < LE1 ? LZ : LE1);
        uint32_t c, lo;
        coords1(tid, c, lo);
        if (KIND == KIND_COL && a.cs_mode == 1u) {   // forward coset transform: x[j] *= s^j, a running product over the registers
            uint32_t tw0 = mont_mul(seed.hi, seed.lo);
#pragma unroll
            for (uint32_t i = 0; i < NZ; ++i) {
                x[i] = mont_mul(x[i], tw0);
                if (i + 1 < NZ) { tw0 = mont_mul_lazy(tw0, a.cs_g); TOYNI_PIN(tw0); }
            }
        }
        if constexpr (LZ == 0) Stg::template stages<LE1, LLO, STEP1_R4>(x, tw1, lo, nullptr, tw3);
        else Stg::template stages_lz<LE1, LLO, LZ, LE1 - 1>(x, tw1, lo, nullptr);
        const uint32_t base = lds_word(c, lo);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) lds[base + i * STRIDE1] = x[i];
    }

    // ---- step 2: the LE2 middle stage bits, in place in LDS ----
    static TOYNI_HD void step2(uint32_t tid, uint32_t* lds, const uint32_t* tw) {
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) {
            uint32_t c, aa, d;
            coords2(tid, g, c, aa, d);
            uint32_t x[E2];
            const uint32_t base = lds_word(c, (aa << LLO) + d);
#pragma unroll
            for (uint32_t j = 0; j < E2; ++j) x[j] = lds[base + j * STRIDE2];
            Stg::template stages<LE2, LE3>(x, tw, d, nullptr);
#pragma unroll
            for (uint32_t j = 0; j < E2; ++j) lds[base + j * STRIDE2] = x[j];
        }
    }

    // streaming shapes: the twiddles of step 2 depend on the thread's d only -- the same for each of its groups and for every tile --
    // so they are read once per workgroup lifetime and stay in registers: w[2^s - 1 + q] = w_{2^(s+LE3+1)}^(d + (q << LE3)), q < 2^s
    struct Tw2 { uint32_t w[E2 - 1 ? E2 - 1 : 1]; };
    static TOYNI_HD Tw2 load_tw2(const PassArgs& a, uint32_t tid) {
        uint32_t c, aa, d;
        coords2(tid, 0u, c, aa, d);
        Tw2 t;
#pragma unroll
        for (int s = 0; s < LE2; ++s) {
#pragma unroll
            for (uint32_t q = 0; q < (1u << s); ++q) t.w[(1u << s) - 1u + q] = a.stage_tw[(1u << (s + LE3)) - 1u + d + (q << LE3)];
        }
        return t;
    }
    static TOYNI_HD void step2_regs(uint32_t tid, uint32_t* lds, const Tw2& tw) {
#pragma unroll
        for (uint32_t g = 0; g < G2; ++g) {
            uint32_t c, aa, d;
            coords2(tid, g, c, aa, d);
            uint32_t x[E2];
            const uint32_t base = lds_word(c, (aa << LLO) + d);
#pragma unroll
            for (uint32_t j = 0; j < E2; ++j) x[j] = lds[base + j * STRIDE2];
#pragma unroll
            for (int s = LE2 - 1; s >= 0; --s) {
#pragma unroll
                for (uint32_t i = 0; i < E2; ++i) {
                    if (i & (1u << s)) continue;
                    const uint32_t w = tw.w[(1u << s) - 1u + (i & ((1u << s) - 1u))];
                    const uint32_t u = x[i], v = x[i + (1u << s)];
                    x[i] = bb_add(u, v);
                    x[i + (1u << s)] = mont_dot_sub(u, v, w, BB_P - w);
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < E2; ++j) lds[base + j * STRIDE2] = x[j];
        }
    }

    // ---- step 3: the LE3 low stage bits (uniform twiddles), inter-pass twiddle / coset output scaling, store ----
    // the factor of output k = (b << LHM) | khi of a group is A * G^b (see Pass::group_twiddle_issue): two-level table lookups,
    // issued ahead of the barriers
    struct SeedsRaw { uint32_t a_lo[G3], a_hi[G3], g_lo, g_hi; };
    struct Seeds { uint32_t a0[G3], g; };
    static TOYNI_HD SeedsRaw seeds_issue(const PassArgs& a, const Tile& t, uint32_t tid) {
        SeedsRaw s{};
#pragma unroll
        for (uint32_t g = 0; g < G3; ++g) {
            uint32_t c, hm;
            coords3(tid, g, c, hm);
            const uint32_t khi = bitrev32(hm, LHM);
            if (KIND == KIND_COL) {
                const uint32_t jcol = (t.col0 + c) >> LQ_;
                const uint32_t mask = (1u << a.tw_lowbits) - 1u;
                const uint32_t ea = jcol * khi;
                s.a_lo[g] = a.tw_lo[ea & mask];
                s.a_hi[g] = a.tw_hi[ea >> a.tw_lowbits];
                if (g == 0) {   // G = w_L^(j' << LHM) depends on the column only, and a thread's groups share it
                    const uint32_t eg = jcol << LHM;
                    s.g_lo = a.tw_lo[eg & mask];
                    s.g_hi = a.tw_hi[eg >> a.tw_lowbits];
                }
            } else if (a.cs_mode == 2u) {
                const uint32_t e0 = t.out0 + (c >> LQ_) + (khi << (a.log_n - LM));
                s.a_lo[g] = a.cs_lo[e0 & ((1u << a.cs_lowbits) - 1u)];
                s.a_hi[g] = a.cs_hi[e0 >> a.cs_lowbits];
            }
        }
        return s;
    }
    static TOYNI_HD Seeds seeds_finish(const PassArgs& a, const SeedsRaw& r) {
        Seeds s{};
        if (KIND == KIND_COL) {
            s.g = mont_mul(r.g_hi, r.g_lo);
#pragma unroll
            for (uint32_t g = 0; g < G3; ++g) {
                s.a0[g] = mont_mul(r.a_hi[g], r.a_lo[g]);
                if (a.scale) s.a0[g] = mont_mul(s.a0[g], a.scale);
            }
        } else if (a.cs_mode == 2u) {
            s.g = a.cs_g;
#pragma unroll
            for (uint32_t g = 0; g < G3; ++g) s.a0[g] = mont_mul(r.a_hi[g], r.a_lo[g]);
        }
        return s;
    }
    static TOYNI_HD void step3(const PassArgs& a, const Tile& t, uint32_t tid, const uint32_t* lds, const Seeds& seeds, const Uniform& uni) {
        const bool twiddled = KIND == KIND_COL || a.cs_mode == 2u;
#pragma unroll
        for (uint32_t g = 0; g < G3; ++g) {
            uint32_t c, hm;
            coords3(tid, g, c, hm);
            uint32_t x[E3];
            const uint32_t base = lds_word(c, hm << LE3);
#pragma unroll
            for (uint32_t j = 0; j < E3; ++j) x[j] = lds[base + j * STRIDE3];
            Stg::template stages<LE3, 0>(x, nullptr, 0u, uni.w);
            const uint32_t khi = bitrev32(hm, LHM);
            const uint32_t off0 = out_offset(a, c, khi) << 2;
            const uint32_t step = (out_offset(a, 0u, 1u << LHM) - out_offset(a, 0u, 0u)) << 2;
            char* obase = reinterpret_cast<char*>(t.out);
            uint32_t tw0 = seeds.a0[g];
#if TOYNI_BUF
            const BufRsrc ws = buf_rsrc(obase);
#endif
#pragma unroll
            for (uint32_t b = 0; b < E3; ++b) {
                uint32_t v = x[cx_bitrev(b, LE3)];
                if (twiddled) {
                    v = mont_mul(v, tw0);
                    if (b + 1 < E3) { tw0 = mont_mul_lazy(tw0, seeds.g); TOYNI_PIN(tw0); }
                }
#if TOYNI_BUF
                if constexpr (STREAM) { stb32<NT_>(ws, off0, b * step, v); continue; }
#endif
                st32<NT_>(reinterpret_cast<uint32_t*>(obase + (uint64_t)b * step), off0, v);
            }
        }
    }

    // whole-tile phases, one thread each (tests/emu runs phase k for every thread before phase k + 1: the pass' two barriers)
    template <int LZ = 0>
    static TOYNI_HD void phase1(const PassArgs& a, uint32_t tile_id, uint32_t tid, uint32_t* lds) {
        const Tile t = tile_of(a, tile_id);
        uint32_t x[E];
        load_tile<LZ>(a, t, tid, x);
        if constexpr (STREAM) step1<LZ>(a, in_seed_issue(a, t, tid), tid, x, lds, tw1s_global(a), tw3s_global(a));
        else step1<LZ>(a, in_seed_issue(a, t, tid), tid, x, lds, tw_global(a) + ((1u << LLO) - E3));
    }
    static TOYNI_HD void phase2(const PassArgs& a, uint32_t tile_id, uint32_t tid, uint32_t* lds) {
        (void)tile_id;
        if constexpr (STREAM) step2_regs(tid, lds, load_tw2(a, tid));
        else step2(tid, lds, tw_global(a));
    }
    static TOYNI_HD void phase3(const PassArgs& a, uint32_t tile_id, uint32_t tid, const uint32_t* lds) {
        const Tile t = tile_of(a, tile_id);
        step3(a, t, tid, lds, seeds_finish(a, seeds_issue(a, t, tid)), load_uniform(a));
    }
};

// ---- single-sweep transforms for n = 2^11 .. 2^15: the whole transform stays in one workgroup's LDS ----
// n = M_a * 1024 with M_a = 2^LA (LA = 1..5).  A tile is R = ROWS / M_a consecutive transforms of the batch = ROWS rows of
// 1024 words = one contiguous block of HBM, read once and written once (the two-pass plan moves it twice).
//   phase A  thread <-> columns j' (= tid + c T, 4-KiB coalesced rows): the M_a-point transforms over j_a for every transform of
//            the tile, in registers, uniform twiddles; times w_n^(j' k_a) (running product of the thread's own w_n^j');
//            row (r, k_a) of the tile is parked in LDS
//   phase B  the 1024-point row transforms, high five stage bits (per-thread twiddles), in place in LDS
//   phase C  low five stage bits (uniform twiddles) and the store: X_r[k_a + M_a k_b].  Lanes run over (k_a, low bits of
//            k_b), so a wave writes whole contiguous runs for every M_a.
// Row layout in LDS as for the row kinds above: word = row * PITCH + hi * 33 + low, PITCH = 1 mod 32 (conflict-free for
// phase A's writes, phase B's strided and phase C's per-row accesses).
constexpr uint32_t BB_MONT_ONE = (uint32_t)((1ull << 32) % BB_P);

struct LdsArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* stage_a;   // packed stage table of the M_a-point transform (uniform twiddles)
    const uint32_t* stage_b;   // packed stage table of the 1024-point transform
    const uint32_t* gtab;      // w_n^j', j' < 1024
    uint32_t scale;            // Montgomery form of n^-1 (inverse transform), 0 = none
    uint64_t batch;            // transforms; the last tile may be ragged
    // coset scaling (PassArgs::cs_*): mode 1 = input x[j] *= s^j, mode 2 = output X[k] *= s^k; cs_g = s^1024 / s^(32 M_a)
    const uint32_t* cs_lo;
    const uint32_t* cs_hi;
    uint32_t cs_lowbits;
    uint32_t cs_mode;
    uint32_t cs_g;
};

template <int LA, int LROWS = 5>
struct LdsPass {
    static_assert(LA >= 1 && LA <= 5 && LROWS >= LA && LROWS <= 5, "n = 2^11 .. 2^15; a tile holds at least one transform");
    // ROWS rows of 1024 words per workgroup: 32 = one 1024-thread workgroup per CU; 16 / 8 = two / four smaller workgroups
    // per CU that drift apart, so one computes while another sits at a barrier or an LDS burst (same waves per CU).
    static constexpr int LOG_N = LA + 10;
    static constexpr uint32_t MA = 1u << LA, ROWS = 1u << LROWS, R = ROWS >> LA, T = ROWS * 32u, E = 32u;
    static constexpr uint32_t NCOL = 1024u / T;                 // columns per thread in phase A (= 32 / ROWS)
    static constexpr uint32_t PITCH = 32u * 33u + 1u;           // 1057 = 1 mod 32
    static constexpr uint32_t LDS_WORDS = ROWS * PITCH;
    static constexpr uint32_t TW1_WORDS = 1024u - 32u;          // stages 5..9 of the 1024-point stage table
    static constexpr uint32_t NU = 16u;                         // uniform twiddles of a 32-point step
    static constexpr uint32_t WG_PER_CU = 32u / ROWS;
    struct Uniform { uint32_t a[MA > 1 ? MA / 2 : 1]; uint32_t b[NU]; };

    static TOYNI_HD Uniform load_uniform(const LdsArgs& g) {
        Uniform u;
#pragma unroll
        for (uint32_t q = 0; q < MA / 2; ++q) u.a[q] = TOYNI_UNIFORM(g.stage_a[MA / 2 - 1u + q]);
#pragma unroll
        for (uint32_t q = 0; q < NU; ++q) u.b[q] = TOYNI_UNIFORM(g.stage_b[NU - 1u + q]);
        return u;
    }
    static TOYNI_HD const uint32_t* tw1_global(const LdsArgs& g) { return g.stage_b + 31u; }
    static TOYNI_HD uint32_t row_word(uint32_t row, uint32_t col) { return row * PITCH + (col >> 5) * 33u + (col & 31u); }

    // phase A, loads: register (c * R + r) * MA + j_a = x_r[j_a * 1024 + j'], j' = tid + c * T (zero beyond the batch)
    static TOYNI_HD void loadA(const LdsArgs& g, uint64_t tile, uint32_t tid, uint32_t (&x)[E]) {
        const uint64_t b0 = tile * R;
        const uint32_t* base = g.in + (b0 << LOG_N);
#pragma unroll
        for (uint32_t c = 0; c < NCOL; ++c) {
#pragma unroll
            for (uint32_t r = 0; r < R; ++r) {
                const bool live = b0 + r < g.batch;   // uniform
#pragma unroll
                for (uint32_t ja = 0; ja < MA; ++ja)
                    x[(c * R + r) * MA + ja] = live ? ld32(base, ((r << LOG_N) + (ja << 10) + tid + c * T) << 2) : 0u;
            }
        }
    }
    // the thread's own constants, held across tiles: w_n^j' for each of its columns.  (The forward coset factor s^j' is
    // looked up per tile instead -- four more live registers push the 8-row shapes over the 128-VGPR budget.)
    struct Seeds { uint32_t g[NCOL]; };
    static TOYNI_HD Seeds seedsA(const LdsArgs& g, uint32_t tid) {
        Seeds s;
#pragma unroll
        for (uint32_t c = 0; c < NCOL; ++c) s.g[c] = g.gtab[tid + c * T];
        return s;
    }
    static TOYNI_HD void phaseA(const LdsArgs& g, uint32_t tid, uint32_t (&x)[E], const Seeds& sd, const Uniform& uni, uint32_t* lds) {
#pragma unroll
        for (uint32_t c = 0; c < NCOL; ++c) {
#pragma unroll
            for (uint32_t r = 0; r < R; ++r) {
                uint32_t (&xr)[MA] = *reinterpret_cast<uint32_t (*)[MA]>(&x[(c * R + r) * MA]);
                if (g.cs_mode == 1u) {   // x[j] *= s^j, j = j_a * 1024 + j'
                    uint32_t j = tid + c * T;
                    TOYNI_PIN(j);        // opaque: keeps the lookup inside the tile loop (hoisted, it would cost NCOL live registers)
                    uint32_t tw = mont_mul(g.cs_hi[j >> g.cs_lowbits], g.cs_lo[j & ((1u << g.cs_lowbits) - 1u)]);
#pragma unroll
                    for (uint32_t ja = 0; ja < MA; ++ja) {
                        xr[ja] = mont_mul(xr[ja], tw);
                        if (ja + 1 < MA) { tw = mont_mul(tw, g.cs_g); TOYNI_PIN(tw); }
                    }
                }
                Pass<KIND_ROW_N, 5, 5, 3>::template stages<LA, 0>(xr, nullptr, 0u, uni.a);
                // times w_n^(j' k_a) (and n^-1 on an inverse): k_a = b sits in register bitrev(b)
                uint32_t tw = g.scale ? g.scale : BB_MONT_ONE;
                if (g.scale) xr[0] = mont_mul(xr[0], tw);
#pragma unroll
                for (uint32_t b = 1; b < MA; ++b) {
                    tw = mont_mul(tw, sd.g[c]);
                    TOYNI_PIN(tw);
                    xr[cx_bitrev(b, LA)] = mont_mul(xr[cx_bitrev(b, LA)], tw);
                }
#pragma unroll
                for (uint32_t b = 0; b < MA; ++b) lds[row_word(r * MA + b, tid + c * T)] = xr[cx_bitrev(b, LA)];
            }
        }
    }
    // phase B: row = tid >> 5, lanes over the low five bits of the column
    static TOYNI_HD void phaseB(uint32_t tid, uint32_t* lds, const uint32_t* tw1) {
        const uint32_t lo = tid & 31u, row = tid >> 5;
        uint32_t x[E];
        const uint32_t base = row * PITCH + lo;
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) x[i] = lds[base + i * 33u];
        Pass<KIND_ROW_N, 5, 5, 3>::template stages<5, 5>(x, tw1, lo, nullptr);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) lds[base + i * 33u] = x[i];
    }
    // phase C: lanes over (k_a, low bits of k_b)
    static TOYNI_HD void phaseC(const LdsArgs& g, uint64_t tile, uint32_t tid, const uint32_t* lds, const Uniform& uni) {
        const uint32_t ka = tid & (MA - 1u), hi_rev = (tid >> LA) & 31u, r = tid >> (LA + 5);
        const uint32_t hi = bitrev32(hi_rev, 5);
        uint32_t x[E];
        const uint32_t base = (r * MA + ka) * PITCH + hi * 33u;
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) x[i] = lds[base + i];
        Pass<KIND_ROW_N, 5, 5, 3>::template stages<5, 0>(x, nullptr, 0u, uni.b);
        const uint64_t b0 = tile * R;
        if (b0 + r >= g.batch) return;
        uint32_t* out = g.out + (b0 << LOG_N);
        const uint32_t k0 = ka + (hi_rev << LA);                 // output index of b = 0; element b sits 32 * M_a further
        const uint32_t off0 = ((r << LOG_N) + k0) << 2;
        if (g.cs_mode == 2u) {                                   // inverse coset transform: X[k] *= s^k
            uint32_t tw = mont_mul(g.cs_hi[k0 >> g.cs_lowbits], g.cs_lo[k0 & ((1u << g.cs_lowbits) - 1u)]);
#pragma unroll
            for (uint32_t b = 0; b < E; ++b) {
                st32(out, off0 + ((b << (5 + LA)) << 2), mont_mul(x[cx_bitrev(b, 5)], tw));
                if (b + 1 < E) { tw = mont_mul(tw, g.cs_g); TOYNI_PIN(tw); }
            }
        } else {
#pragma unroll
            for (uint32_t b = 0; b < E; ++b) st32(out, off0 + ((b << (5 + LA)) << 2), x[cx_bitrev(b, 5)]);
        }
    }
};

// ---- n = 2^11 in ONE sweep, ONE WAVE per transform (round 5) --------------------------------------------------------------------
// A 2048-point transform is 32 elements per lane of one wave.  The three register steps of the streaming three-step shape
// (Pass3<..., 5, 3, 3, ...>) then exchange data only WITHIN the wave, so a row needs no workgroup barrier at all: a wave loads its row
// (contiguous, 256 bytes per instruction), runs steps 1 / 2 / 3 through its private 2112-word slice of LDS (LDS operations of one wave
// execute in order) and stores the row in natural order (contiguous again).  The 16 waves of a workgroup share nothing but the
// step-1 twiddle slices; each walks its own rows with the next row's loads in flight behind the current row's stores.
//   position r = (a << 6) | (b << 3) | d   a: 5 bits (step 1, registers), b: 3 bits (step 2), d: 3 bits (step 3)
//   LDS word of position r: r + (r >> 5)  (one pad word per 32), which makes every access of every step conflict-free:
//     step 1  lane = lo (b, d), registers over a        : words lo + (lo >> 5) + 66 i                  -- consecutive lanes, consecutive words
//     step 2  lanes over (a & 15, d & 3), registers over b: words 66 a + d + {0, 8, 16, 24, 33, 41, 49, 57}   -- bank 2 a + d: 32 different ones
//     step 3  lane l <-> hm = (a, b) = 4 l + rev2(g), registers over d: words 33 l + 8 rev2(g) + j         -- bank l
//   output sub-index of register j of lane l, group g in step 3: k = rev3(j) << 8 | (g << 6) | rev6(l): a wave's store covers 64
//   consecutive words (in a permuted lane order, which costs nothing).
// Replaces, for n = 2^11, the single-sweep LdsPass kernel (3.6e11 elements/s: 61 lane-operations per element at 56 % VALU
// utilisation, three workgroup barriers per tile).
struct Row2048 {
    static constexpr int LM = 11;
    static constexpr uint32_t M = 2048, E = 32, WAVES = 16, T = 64 * WAVES;
    static constexpr uint32_t ROW_WORDS = M + (M >> 5);
    static constexpr uint32_t LDS_WORDS = WAVES * ROW_WORDS;
    static constexpr uint32_t TW1_WORDS = M - 64u;      // stages 6 .. 10 of the packed 2048-point stage table
    static constexpr uint32_t TW3_WORDS = M - 128u;     // blocks 7 .. 10 of its radix-4 companion
    using Stg = Pass<KIND_ROW_N, 5, 5, 3>;
    static TOYNI_HD const uint32_t* tw1_global(const PassArgs& a) { return a.stage_tw + 63u; }
    static TOYNI_HD const uint32_t* tw3_global(const PassArgs& a) { return a.stage_tw3 + 126u; }
    static TOYNI_HD uint32_t word(uint32_t r) { return r + (r >> 5); }

    static constexpr uint32_t TW2_WORDS = 56u;          // stages 3 .. 5 (step 2): words [7, 63) of the table
    static TOYNI_HD const uint32_t* tw2_global(const PassArgs& a) { return a.stage_tw + 7u; }
    // (the step-2 twiddles live in LDS too and the coset seeds are looked up per row: held in registers for the whole launch they
    // pushed the kernel to 128 VGPRs + 92 bytes of scratch)
    struct Consts { uint32_t uni[4]; };   // w_8^q, wave-uniform (SGPRs)
    static TOYNI_HD uint32_t d_of(uint32_t l, uint32_t g) { return ((l >> 4) & 3u) | ((g >> 1) << 2); }
    static TOYNI_HD uint32_t a_of(uint32_t l, uint32_t g) { return (l & 15u) | ((g & 1u) << 4); }
    static TOYNI_HD uint32_t kappa_of(uint32_t l, uint32_t g) { return (g << 6) | bitrev32(l, 6); }
    static TOYNI_HD Consts consts(const PassArgs& a) {
        Consts c{};
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) c.uni[q] = TOYNI_UNIFORM(a.stage_tw[3u + q]);
        return c;
    }
    template <bool NT>
    static TOYNI_HD void load_row(const PassArgs& a, uint64_t row, uint32_t l, uint32_t (&x)[E]) {
        const uint32_t* base = a.in + (row << LM);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) x[i] = ld32<NT>(base + i * 64u, l << 2);
    }
    // step 1: coset input scaling, the five high stage bits (radix 4, twiddle slices in LDS), park in the wave's LDS slice
    static TOYNI_HD void step1(const PassArgs& a, uint32_t l, uint32_t (&x)[E], uint32_t* row_lds, const uint32_t* tw1, const uint32_t* tw3) {
        if (a.cs_mode == 1u) {   // x[j] *= s^j, j = l + 64 i: the lane's seed s^l, then a running product with s^64
            uint32_t tw = mont_mul(a.cs_hi[l >> a.cs_lowbits], a.cs_lo[l & ((1u << a.cs_lowbits) - 1u)]);
#pragma unroll
            for (uint32_t i = 0; i < E; ++i) {
                x[i] = mont_mul(x[i], tw);
                if (i + 1 < E) { tw = mont_mul_lazy(tw, a.cs_g); TOYNI_PIN(tw); }
            }
        }
        Stg::template stages<5, 6, true>(x, tw1, l, nullptr, tw3);
        const uint32_t base = l + (l >> 5);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) row_lds[base + i * 66u] = x[i];
    }
    // step 2: the three middle stage bits, in place; twiddles from registers
    // tw2: words [7, 63) of the packed stage table (stages 3 .. 5): entry (1 << (s + 3)) - 8 + d + 8 q = w_{2^(s+4)}^(d + 8 q)
    static TOYNI_HD void step2(uint32_t l, uint32_t* row_lds, const uint32_t* tw2) {
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t d = d_of(l, g);
            const uint32_t base = 66u * a_of(l, g) + d;
            uint32_t tw[7];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (uint32_t q = 0; q < (1u << s); ++q) tw[(1u << s) - 1u + q] = tw2[(1u << (s + 3)) - 8u + d + (q << 3)];
            uint32_t x[8];
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) x[b] = row_lds[base + 8u * b + (b >> 2)];
#pragma unroll
            for (int s = 2; s >= 0; --s) {
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    if (i & (1u << s)) continue;
                    const uint32_t w = tw[(1u << s) - 1u + (i & ((1u << s) - 1u))];
                    const uint32_t u = x[i], v = x[i + (1u << s)];
                    x[i] = bb_add(u, v);
                    x[i + (1u << s)] = mont_dot_sub(u, v, w, BB_P - w);
                }
            }
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) row_lds[base + 8u * b + (b >> 2)] = x[b];
        }
    }
    // step 3: the three low stage bits (uniform twiddles), output scaling, natural-order store
    template <bool NT>
    static TOYNI_HD void step3(const PassArgs& a, const Consts& c, uint64_t row, uint32_t l, const uint32_t* row_lds) {
        uint32_t* out = a.out + (row << LM);
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t base = 33u * l + 8u * cx_bitrev(g, 2);
            uint32_t x[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) x[j] = row_lds[base + j];
            Stg::template stages<3, 0>(x, nullptr, 0u, c.uni);
            const uint32_t kappa = kappa_of(l, g);
            const uint32_t off0 = kappa << 2;
            uint32_t tw = 0u;
            if (a.cs_mode == 2u) {   // X[k] *= s^k, k = (b << 8) | kappa: seed s^kappa (times n^-1), then a running product with s^256
                tw = mont_mul(a.cs_hi[kappa >> a.cs_lowbits], a.cs_lo[kappa & ((1u << a.cs_lowbits) - 1u)]);
                if (a.scale) tw = mont_mul(tw, a.scale);
            }
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {   // store b carries output k = (b << 8) | kappa, held by register rev3(b)
                uint32_t v = x[cx_bitrev(b, 3)];
                if (a.cs_mode == 2u) {
                    v = mont_mul(v, tw);
                    if (b + 1 < 8) { tw = mont_mul_lazy(tw, a.cs_g); TOYNI_PIN(tw); }
                } else if (a.scale) {
                    v = mont_mul(v, a.scale);
                }
                st32<NT>(out + b * 256u, off0, v);
            }
        }
    }
};

// ---- n = 2^12 in ONE sweep, TWO waves per transform (round 5) ------------------------------------------------------------------
// The same three register steps for a 4096-point row: 128 threads (two waves) per transform, eight transforms per 1024-thread
// workgroup, contiguous loads and stores.  Two waves share a row, so the steps are separated by workgroup barriers again (an
// s_barrier cannot name a pair of waves), and the workgroup walks TILES of eight rows in lockstep.
//   position r = (a << 7) | (b << 3) | d   a: 5 bits (step 1, registers), b: 4 bits (step 2), d: 3 bits (step 3); LDS word r + (r >> 5)
//     step 1  thread tau = (b, d), registers over a          : words tau + (tau >> 5) + 132 i
//     step 2  lanes over (a & 7, d & 3), registers over b    : words 132 a + d + 8 b + (b >> 2)        -- bank 4 a + d: 32 different ones
//     step 3  u = 2 (tau & 63) + (tau >> 6), hm = 4 u + rev2(g), registers over d: words 33 u + 8 rev2(g) + j  -- bank 2 t + w: two-way
//   output sub-index in step 3: k = rev3(j) << 9 | (g << 7) | (wave of the pair << 6) | rev6(lane): 64 consecutive words per store.
// The step-1 stages are radix 2 (the radix-4 companion slice would not fit next to eight rows: 151.5 KiB as it is).
struct Row4096 {
    static constexpr int LM = 12;
    static constexpr uint32_t M = 4096, E = 32, ROWS = 8, T = 1024;
    static constexpr uint32_t ROW_WORDS = M + (M >> 5);
    static constexpr uint32_t LDS_WORDS = ROWS * ROW_WORDS;
    static constexpr uint32_t TW1_WORDS = M - 128u;     // stages 7 .. 11 of the packed 4096-point stage table
    static constexpr uint32_t TW2_WORDS = 120u;         // stages 3 .. 6 (step 2): words [7, 127)
    using Stg = Pass<KIND_ROW_N, 5, 5, 3>;
    static TOYNI_HD const uint32_t* tw1_global(const PassArgs& a) { return a.stage_tw + 127u; }
    static TOYNI_HD const uint32_t* tw2_global(const PassArgs& a) { return a.stage_tw + 7u; }
    struct Consts { uint32_t uni[4]; };   // w_8^q, wave-uniform (SGPRs)
    static TOYNI_HD Consts consts(const PassArgs& a) {
        Consts c{};
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) c.uni[q] = TOYNI_UNIFORM(a.stage_tw[3u + q]);
        return c;
    }
    static TOYNI_HD uint32_t kappa_of(uint32_t tau, uint32_t g) { return (g << 7) | ((tau >> 6) << 6) | bitrev32(tau & 63u, 6); }
    template <bool NT>
    static TOYNI_HD void load_row(const PassArgs& a, uint64_t row, uint32_t tau, uint32_t (&x)[E]) {
        const uint32_t* base = a.in + (row << LM);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) x[i] = ld32<NT>(base + i * 128u, tau << 2);
    }
    static TOYNI_HD void step1(const PassArgs& a, uint32_t tau, uint32_t (&x)[E], uint32_t* row_lds, const uint32_t* tw1) {
        if (a.cs_mode == 1u) {   // x[j] *= s^j, j = tau + 128 i
            uint32_t tw = mont_mul(a.cs_hi[tau >> a.cs_lowbits], a.cs_lo[tau & ((1u << a.cs_lowbits) - 1u)]);
#pragma unroll
            for (uint32_t i = 0; i < E; ++i) {
                x[i] = mont_mul(x[i], tw);
                if (i + 1 < E) { tw = mont_mul_lazy(tw, a.cs_g); TOYNI_PIN(tw); }
            }
        }
        Stg::template stages<5, 7>(x, tw1, tau, nullptr);
        const uint32_t base = tau + (tau >> 5);
#pragma unroll
        for (uint32_t i = 0; i < E; ++i) row_lds[base + i * 132u] = x[i];
    }
    // tw2: words [7, 127) of the packed stage table: entry (1 << (s + 3)) - 8 + d + 8 q = w_{2^(s+4)}^(d + 8 q)
    static TOYNI_HD void step2(uint32_t tau, uint32_t* row_lds, const uint32_t* tw2) {
        const uint32_t av = (tau & 7u) | (((tau >> 5) & 3u) << 3);
#pragma unroll
        for (uint32_t g = 0; g < 2; ++g) {
            const uint32_t d = ((tau >> 3) & 3u) | (g << 2);
            const uint32_t base = 132u * av + d;
            uint32_t x[16];
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) x[b] = row_lds[base + 8u * b + (b >> 2)];
#pragma unroll
            for (int s = 3; s >= 0; --s) {
                uint32_t tw[8];
#pragma unroll
                for (uint32_t q = 0; q < (1u << s); ++q) tw[q] = tw2[(1u << (s + 3)) - 8u + d + (q << 3)];
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) {
                    if (i & (1u << s)) continue;
                    const uint32_t w = tw[i & ((1u << s) - 1u)];
                    const uint32_t u = x[i], v = x[i + (1u << s)];
                    x[i] = bb_add(u, v);
                    x[i + (1u << s)] = mont_dot_sub(u, v, w, BB_P - w);
                }
            }
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) row_lds[base + 8u * b + (b >> 2)] = x[b];
        }
    }
    template <bool NT>
    static TOYNI_HD void step3(const PassArgs& a, const Consts& c, uint64_t row, uint32_t tau, const uint32_t* row_lds) {
        uint32_t* out = a.out + (row << LM);
        const uint32_t u = ((tau & 63u) << 1) | (tau >> 6);
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t base = 33u * u + 8u * cx_bitrev(g, 2);
            uint32_t x[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) x[j] = row_lds[base + j];
            Stg::template stages<3, 0>(x, nullptr, 0u, c.uni);
            const uint32_t kappa = kappa_of(tau, g);
            const uint32_t off0 = kappa << 2;
            uint32_t tw = 0u;
            if (a.cs_mode == 2u) {   // X[k] *= s^k, k = (b << 9) | kappa
                tw = mont_mul(a.cs_hi[kappa >> a.cs_lowbits], a.cs_lo[kappa & ((1u << a.cs_lowbits) - 1u)]);
                if (a.scale) tw = mont_mul(tw, a.scale);
            }
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                uint32_t v = x[cx_bitrev(b, 3)];
                if (a.cs_mode == 2u) {
                    v = mont_mul(v, tw);
                    if (b + 1 < 8) { tw = mont_mul_lazy(tw, a.cs_g); TOYNI_PIN(tw); }
                } else if (a.scale) {
                    v = mont_mul(v, a.scale);
                }
                st32<NT>(out + b * 512u, off0, v);
            }
        }
    }
};

// ---- u64 <-> u32 edge of the reference-shaped entry points (src/ntt.rs:233: &mut [BabyBear] as *mut u64) ----
// narrow also reduces mod p, so a non-canonical u64 behaves like BabyBear::new (src/babybear.rs:26-30)
TOYNI_HD uint32_t narrow_u64(uint64_t v) { return (uint32_t)(v % BB_P); }

// ---- re-layout around the one exchange of a multi-device transform (toyni_ntt_slab_relayout_device) ----
// rows = this rank's k_1 block, parts = ranks, W = S_1 / parts columns per piece.
//   forward (after the exchange) : in [parts][rows][W] -> out [rows][parts][W]        (pieces -> contiguous rows)
//   inverse (before the exchange): in [rows][parts][W] -> out [parts][rows][W], times w_n^-(k1 j'), k1 = row0 + r,
//                                  j' = g W + w -- the twiddle between the row transforms and the column transforms
struct RelayoutArgs {
    const uint32_t* in;
    uint32_t* out;
    uint32_t log_rows, log_parts, log_w;
    uint32_t inverse, row0;
    const uint32_t* lo;   // w_n^-x two-level table (inverse only)
    const uint32_t* hi;
    uint32_t lowbits;
};
// source index and twiddle exponent of OUTPUT element o (consecutive o within a piece map to consecutive sources)
TOYNI_HD uint64_t relayout_src(const RelayoutArgs& a, uint64_t o, uint32_t& exponent) {
    const uint32_t w = (uint32_t)o & ((1u << a.log_w) - 1u);
    uint32_t r, g;
    if (!a.inverse) {
        g = (uint32_t)(o >> a.log_w) & ((1u << a.log_parts) - 1u);
        r = (uint32_t)(o >> (a.log_w + a.log_parts));
        exponent = 0u;
        return ((((uint64_t)g << a.log_rows) + r) << a.log_w) + w;
    }
    r = (uint32_t)(o >> a.log_w) & ((1u << a.log_rows) - 1u);
    g = (uint32_t)(o >> (a.log_w + a.log_rows));
    exponent = (a.row0 + r) * ((g << a.log_w) + w);  // k1 j' < n <= 2^27
    return ((((uint64_t)r << a.log_parts) + g) << a.log_w) + w;
}
TOYNI_HD uint32_t relayout_value(const RelayoutArgs& a, uint32_t v, uint32_t exponent) {
    if (!a.inverse) return v;
    const uint32_t tw = mont_mul(a.hi[exponent >> a.lowbits], a.lo[exponent & ((1u << a.lowbits) - 1u)]);
    return mont_mul(v, tw);
}

// ---- FRI pairwise fold (src/math/fri.rs:27-48), structured-domain form ----
// Layer points are x_i = x0 * w_m^i (prover: src/fibonacci.rs:214,228-231), so
//   out[i] = (a+b)/2 + (a-b) * [ (beta / (2 x0)) * w_m^-i ],   a = evals[i], b = evals[i + m/2].
// w_m^-i = winv_N^(i << s), s = log2(N / m), comes from the ctx' two-level inverse-root table through a SubDomain view.
//
// SubDomain: the order-m subgroup inside a context's order-N domain table (lo[x] = w^x for x < 2^L, hi[y] = w^(y << L)):
//   w^(i << s) = hi[(i >> rsh) << lsh] * lo_s[i & lo_mask]
// For s = 0 lo_s is the table's own low level.  For 0 < s < L the exponent's low part (i << s) & (2^L - 1) would walk that level at a
// stride of 2^s words -- a wave's 64 lookups on 64 different cache lines; measured: the structured fold of a 2^24 layer on a 2^27
// context took 28 us against 17 us on a 2^24 context -- so the plan holds a COMPACT low level per s (lo_s[x] = w^(x << s),
// x < 2^(L-s): 2^L words over all s together) and consecutive points read consecutive words again.  For s >= L the low part is 0.
struct SubDomain {
    const uint32_t* lo;
    const uint32_t* hi;
    uint32_t lo_mask, rsh, lsh;
};
TOYNI_HD uint32_t subdomain_mont(const SubDomain& d, uint32_t i) {   // Montgomery form of w^(i << s)
    return mont_mul(d.hi[(i >> d.rsh) << d.lsh], d.lo[i & d.lo_mask]);
}

struct FoldArgs {
    const uint32_t* evals;
    uint32_t* out;
    SubDomain dom;            // winv_N^(i << s) for i < m (Montgomery)
    uint32_t coef;            // Montgomery form of beta / (2 x0)
    uint64_t half;            // m / 2
    uint32_t step;            // Montgomery form of w_m^-1: the factor between the points of two consecutive outputs (fold_quad)
};

TOYNI_HD uint32_t fold_one(const FoldArgs& f, uint64_t i, uint32_t a, uint32_t b) {
    const uint32_t w = subdomain_mont(f.dom, (uint32_t)i);   // Montgomery form of w_m^-i
    const uint32_t cw = mont_mul(w, f.coef);                 // Montgomery form of coef * w_m^-i
    const uint32_t avg = bb_halve(bb_add(a, b));
    return bb_add(avg, mont_mul(bb_sub_lazy(a, b), cw));
}

// Four consecutive outputs i0 .. i0 + 3 (i0 a multiple of 4): ONE table lookup pair for the first point, the other three by a running
// product with w_m^-1 -- 2 gathers and 9 Montgomery products per four outputs where fold_one spends 8 and 12.  The sweep is
// memory-bound, but the eight 4-byte gathers per 16-byte load pair kept the address path as busy as the stream itself.
TOYNI_HD void fold_quad(const FoldArgs& f, uint64_t i0, const uint32_t (&a)[4], const uint32_t (&b)[4], uint32_t (&r)[4]) {
    const uint32_t w = subdomain_mont(f.dom, (uint32_t)i0);
    uint32_t cw = mont_mul(w, f.coef);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = bb_add(bb_halve(bb_add(a[j], b[j])), mont_mul(bb_sub_lazy(a[j], b[j]), cw));
        if (j < 3) cw = mont_mul(cw, f.step);
    }
}

// ---- Ext = F_p[X]/(X^4 - 11) (src/ext.rs), only what fri_fold_ext needs ----
struct Ext4 { uint32_t c[4]; };
// The fixed right-hand factor of an Ext product, prepared once: coordinates in Montgomery form plus 11 * coordinates
// (X^4 = 11 folds the high half of the schoolbook product back, src/ext.rs:178-192).
struct ExtFactor { uint32_t b[4], b11[4]; };
inline ExtFactor ext_factor_host(const uint32_t beta[4]) {
    ExtFactor f;
    for (int k = 0; k < 4; ++k) { f.b[k] = to_mont_host(beta[k]); f.b11[k] = to_mont_host(bb_mul_host(beta[k], 11u)); }
    return f;
}
// a (plain canonical coordinates) * f  ->  plain canonical coordinates; four 2-term dot products per coordinate pair
TOYNI_HD Ext4 ext_mul(const Ext4& a, const ExtFactor& f) {
    Ext4 r;
    r.c[0] = bb_add(mont_dot2(a.c[0], f.b[0], a.c[1], f.b11[3]), mont_dot2(a.c[2], f.b11[2], a.c[3], f.b11[1]));
    r.c[1] = bb_add(mont_dot2(a.c[0], f.b[1], a.c[1], f.b[0]), mont_dot2(a.c[2], f.b11[3], a.c[3], f.b11[2]));
    r.c[2] = bb_add(mont_dot2(a.c[0], f.b[2], a.c[1], f.b[1]), mont_dot2(a.c[2], f.b[0], a.c[3], f.b11[3]));
    r.c[3] = bb_add(mont_dot2(a.c[0], f.b[3], a.c[1], f.b[2]), mont_dot2(a.c[2], f.b[1], a.c[3], f.b[0]));
    return r;
}

// fri_fold_ext (src/math/fri.rs:7-25): out = (a+b)/2 + ((a-b)/2 * beta) * x^-1 coordinate-wise avg/diff, Ext product with beta.
// `scaleR` = Montgomery form of the base-field factor applied to (a-b): x^-1 (times whatever the caller folded in);
// `f` = ExtFactor of beta/2.
TOYNI_HD Ext4 fold_ext_one(const Ext4& a, const Ext4& b, uint32_t scaleR, const ExtFactor& f) {
    Ext4 d, avg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        avg.c[k] = bb_halve(bb_add(a.c[k], b.c[k]));
        d.c[k] = mont_mul(bb_sub_lazy(a.c[k], b.c[k]), scaleR);
    }
    const Ext4 p = ext_mul(d, f);
    Ext4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.c[k] = bb_add(avg.c[k], p.c[k]);
    return r;
}

// ---- FRI fold with explicit points (the reference's signature fri_fold(evals, xs, beta)) ----
// x^-1 by Fermat like src/babybear.rs:111-114, shared over B elements with Montgomery's trick.
TOYNI_HD uint32_t bb_mul_plain(uint32_t a, uint32_t b) { return mont_mul(a, to_mont(b)); }

// cw[j] = f * x_j^-1 * R (the Montgomery form of f / x_j, canonical) for B PLAIN canonical points behind ONE inversion; a zero point
// rides along as 1 and gets 0 = f * pow(0, p - 2), what BabyBear::inverse computes without its zero assert (src/babybear.rs:111-114).
// Round 4: no conversion of the points and no separate product with f.  With M(a, b) = a b / R and the points left in plain form the
// powers of R that pile up are the same for every j, and they cancel:
//   pre_j = R prod_{i<j} x_i / R^j                (pre_0 = R is never multiplied: pre_1 = x_0)
//   inv   = R^2 / pre_B                           (pre_B read as a Montgomery form: mont_inv_chain)
//   I_{B-1} = M(inv, f),  I_{j-1} = M(I_j, x_j)   ->  M(I_j, pre_j) = R f / x_j,  and M(I_0, pre_0) = I_0
// so a point costs 3 products (prefix, back-substitution, cw) + 41 / B of the chain, and the I chain stays in [0, 2p) (mont_mul_lazy:
// the left operand may be any u32): 13 instructions per point + 205 / B, where round 3's form (points to Montgomery form, (beta/2) as a
// fourth product, the 60-product ladder) spent 30 + 300 / B.
template <int B>
TOYNI_HD void batch_inverse_scaled(const uint32_t (&x)[B], uint32_t f, uint32_t (&cw)[B]) {
    uint32_t xn[B], pre[B];
#pragma unroll
    for (int j = 0; j < B; ++j) xn[j] = x[j] > 1u ? x[j] : 1u;
    pre[0] = BB_R1;
    uint32_t acc = xn[0];
#pragma unroll
    for (int j = 1; j < B; ++j) {
        pre[j] = acc;
        acc = mont_mul(acc, xn[j]);
    }
    uint32_t inv = mont_mul_lazy(mont_inv_chain(acc), f);
#pragma unroll
    for (int j = B - 1; j >= 1; --j) {
        cw[j] = mont_mul(inv, pre[j]);
        inv = mont_mul_lazy(inv, xn[j]);
    }
    cw[0] = bb_reduce_2p(inv);
#pragma unroll
    for (int j = 0; j < B; ++j) cw[j] &= 0u - (x[j] < 1u ? x[j] : 1u);   // all ones unless x_j = 0
}

// B outputs of the explicit-point fold behind one inversion: r[j] = (a_j + b_j)/2 + (a_j - b_j) (beta/2) / x_j; beta_half = beta / 2, PLAIN
template <int B>
TOYNI_HD void fold_xs_batch(const uint32_t (&x)[B], const uint32_t (&a)[B], const uint32_t (&b)[B], uint32_t beta_half, uint32_t (&r)[B]) {
    uint32_t cw[B];
    batch_inverse_scaled<B>(x, beta_half, cw);
#pragma unroll
    for (int j = 0; j < B; ++j) r[j] = bb_add(bb_halve(bb_add(a[j], b[j])), mont_mul(bb_sub_lazy(a[j], b[j]), cw[j]));
}

}  // namespace toyni
