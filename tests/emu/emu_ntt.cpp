// CPU stepping of the gfx950 kernel bodies (toyni_amd/csrc/ntt_kernels.hpp) against the oracle.
//
// TEST INFRASTRUCTURE.  The shipped library has no CPU path; this harness exists so that the index
// algebra of every pass shape (tile maps, LDS swizzles, twiddle tables, digit reversal, pass
// sequencing) is proven on the CPU before a GPU minute is spent.  A workgroup is emulated by
// running phase1 for every thread id, then phase2 for every thread id (the one barrier of a pass).
//
// Build: g++ -O2 -std=c++17 -I toyni_amd/csrc tests/emu/emu_ntt.cpp oracle/toyni_oracle.c  (see tests/test_emu.py)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ntt_plan.hpp"
#include "merkle_kernels.hpp"
#include "prover_kernels.hpp"

extern "C" {
uint64_t orc_bb_mul(uint64_t, uint64_t);
uint64_t orc_bb_add(uint64_t, uint64_t);
uint64_t orc_bb_sub(uint64_t, uint64_t);
uint64_t orc_bb_inverse(uint64_t);
uint64_t orc_bb_pow(uint64_t, uint64_t);
int orc_ntt_canonical(uint64_t*, size_t);
int orc_intt_canonical(uint64_t*, size_t);
void orc_fill_splitmix(uint64_t*, size_t, uint64_t);
void orc_fill_pattern_7i3(uint64_t*, size_t);
int orc_fri_fold(uint64_t*, const uint64_t*, size_t, const uint64_t*, uint64_t);
int orc_domain_elements(uint64_t*, size_t, uint64_t);
int orc_fri_fold_ext(uint64_t*, const uint64_t*, size_t, const uint64_t*, const uint64_t*);
int orc_domain_fft(uint64_t*, size_t, const uint64_t*, size_t, uint64_t);
void orc_merkle_commit_values(uint8_t*, const uint64_t*, const uint8_t*, size_t);
size_t orc_merkle_total_digests(size_t);
int orc_domain_ifft(uint64_t*, size_t, uint64_t);
uint64_t orc_poly_eval(const uint64_t*, size_t, uint64_t);
int orc_fib_quotient(uint64_t*, uint64_t*, const uint64_t*, size_t, size_t, uint64_t);
int orc_fib_deep(uint64_t*, const uint64_t*, const uint64_t*, size_t, size_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
int orc_merkle_get_proof(uint8_t*, uint8_t*, const uint8_t*, size_t, size_t);
uint64_t orc_bb_root_of_unity(uint32_t);
}

using namespace toyni;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// one tile of a pass, phase by phase: every thread runs phase k before any thread runs phase k + 1 (the pass' barriers)
static unsigned long long tiles_by_steps[4] = {0, 0, 0, 0};
static bool latency_plan = false;   // "Q1": n = 2^21 / 2^22 through their two-pass plans (2048-point three-step shapes)
template <class P, int LZ = 0>
static void emu_tile(const PassArgs& a, uint32_t b, uint32_t* lds) {
    ++tiles_by_steps[P::STEPS];
    if constexpr (P::STEPS == 3) {
        if constexpr (P::WAVE_LOCAL2) {
            // the streaming row shapes have NO barrier between steps 1 and 2 (a wave's step 2 reads only what the same wave parked in
            // step 1): stepped wave by wave, so a read of another wave's row would see the 0xDEADBEEF fill / the previous tile's data
            for (uint32_t w = 0; w < P::T / 64; ++w) {
                for (uint32_t tid = 64 * w; tid < 64 * (w + 1); ++tid) P::template phase1<LZ>(a, b, tid, lds);
                for (uint32_t tid = 64 * w; tid < 64 * (w + 1); ++tid) P::phase2(a, b, tid, lds);
            }
            for (uint32_t tid = 0; tid < P::T; ++tid) P::phase3(a, b, tid, lds);
            return;
        }
    }
    for (uint32_t tid = 0; tid < P::T; ++tid) P::template phase1<LZ>(a, b, tid, lds);
    if constexpr (P::STEPS >= 2) {
        for (uint32_t tid = 0; tid < P::T; ++tid) P::phase2(a, b, tid, lds);
    }
    if constexpr (P::STEPS >= 3) {
        for (uint32_t tid = 0; tid < P::T; ++tid) P::phase3(a, b, tid, lds);
    }
}

static uint64_t lde_batch = 2;      // "bN": vectors per low-degree-extension case ("lLOGxZ")
static bool use_row2048 = false;   // "R1": n = 2^11 through the one-wave-per-transform kernel (Row2048)
static int lds_rows = 5;      // rows per workgroup of the single-sweep kernel (2^lds_rows)
static bool use_lds = true;   // sizes 2^11 .. 2^15 have two executors: the single-sweep kernel and the two-pass plan
// LQ > 0: the interleaved (Ext, AoS) passes -- `batch` counts base-field transforms, 2^LQ of them interleaved word by word
template <int LQ = 0>
static void emu_transform(const NttPlan& plan, bool inverse, const uint32_t* src, uint32_t* work, uint32_t* dst, uint64_t batch, uint32_t shift = 1,
                          int lde_log = 0) {
    const std::vector<uint32_t>& blob = inverse ? plan.inv : plan.fwd;
    std::vector<uint32_t> cblob;
    CosetTables cs;
    if (shift != 1 && plan.log_n > 0) {
        uint32_t lo_off, hi_off;
        cs.s = inverse ? bb_inv_host(shift) : shift;
        append_two_level(cblob, plan.log_n, cs.s, 1u, lo_off, hi_off, cs.lowbits);
        cs.lo = cblob.data() + lo_off;
        cs.hi = cblob.data() + hi_off;
    }
    if (plan.log_n == 0 && src != dst) std::memcpy(dst, src, batch * sizeof(uint32_t));
    if (LQ == 0 && use_row2048 && plan.log_n == 12 && !lde_log) {   // two waves per transform: the pair steps through its row, a barrier between steps
        bool okr = row2048_transform(plan, blob.data(), inverse, src, dst, batch, [&](const PassArgs& a, uint64_t rows) {
            using R = Row4096;
            std::vector<uint32_t> row_lds(R::ROW_WORDS, 0xDEADBEEFu);
            std::vector<uint32_t> regs(128 * R::E);
            for (uint64_t row = 0; row < rows; ++row) {
                for (uint32_t t = 0; t < 128; ++t) R::load_row<false>(a, row, t, *reinterpret_cast<uint32_t (*)[R::E]>(&regs[(size_t)t * R::E]));
                for (uint32_t t = 0; t < 128; ++t) R::step1(a, t, *reinterpret_cast<uint32_t (*)[R::E]>(&regs[(size_t)t * R::E]), row_lds.data(), R::tw1_global(a));
                for (uint32_t t = 0; t < 128; ++t) R::step2(t, row_lds.data(), R::tw2_global(a));
                for (uint32_t t = 0; t < 128; ++t) R::step3<false>(a, R::consts(a), row, t, row_lds.data());
            }
        }, cs);
        CHECK(okr, "row4096 transform rejected log_n=%d", plan.log_n);
        return;
    }
    if (LQ == 0 && use_row2048 && plan.log_n == 11 && !lde_log) {   // one wave per transform: stepped wave by wave, lane by lane within a step
        bool okr = row2048_transform(plan, blob.data(), inverse, src, dst, batch, [&](const PassArgs& a, uint64_t rows) {
            using R = Row2048;
            std::vector<uint32_t> row_lds(R::ROW_WORDS, 0xDEADBEEFu);   // exactly one wave's slice: an overrun is an ASan error
            std::vector<uint32_t> regs(64 * R::E);
            for (uint64_t row = 0; row < rows; ++row) {
                for (uint32_t l = 0; l < 64; ++l) {      // every load of the row before any store (in-place transforms)
                    uint32_t (&x)[R::E] = *reinterpret_cast<uint32_t (*)[R::E]>(&regs[(size_t)l * R::E]);
                    R::load_row<false>(a, row, l, x);
                }
                for (uint32_t l = 0; l < 64; ++l) {
                    uint32_t (&x)[R::E] = *reinterpret_cast<uint32_t (*)[R::E]>(&regs[(size_t)l * R::E]);
                    R::step1(a, l, x, row_lds.data(), R::tw1_global(a), R::tw3_global(a));
                }
                for (uint32_t l = 0; l < 64; ++l) R::step2(l, row_lds.data(), R::tw2_global(a));
                for (uint32_t l = 0; l < 64; ++l) R::step3<false>(a, R::consts(a), row, l, row_lds.data());
            }
        }, cs);
        CHECK(okr, "row2048 transform rejected log_n=%d", plan.log_n);
        return;
    }
    if (LQ == 0 && use_lds && plan.lds_la && !lde_log) {      // n = 2^11 .. 2^15: the single-sweep kernel, phase by phase (two barriers)
        bool okl = lds_transform(plan, blob.data(), inverse, src, dst, batch, [&](auto pass, const LdsArgs& g, uint64_t ntiles) {
            using L = decltype(pass);
            std::vector<uint32_t> lds(L::LDS_WORDS, 0xDEADBEEFu);
            std::vector<uint32_t> regs((size_t)L::T * L::E);
            for (uint64_t tile = 0; tile < ntiles; ++tile) {
                for (uint32_t tid = 0; tid < L::T; ++tid) {
                    uint32_t (&x)[L::E] = *reinterpret_cast<uint32_t (*)[L::E]>(&regs[(size_t)tid * L::E]);
                    L::loadA(g, tile, tid, x);      // every load of the tile before any store (in-place transforms)
                }
                for (uint32_t tid = 0; tid < L::T; ++tid) {
                    uint32_t (&x)[L::E] = *reinterpret_cast<uint32_t (*)[L::E]>(&regs[(size_t)tid * L::E]);
                    L::phaseA(g, tid, x, L::seedsA(g, tid), L::load_uniform(g), lds.data());
                }
                for (uint32_t tid = 0; tid < L::T; ++tid) L::phaseB(tid, lds.data(), L::tw1_global(g));
                for (uint32_t tid = 0; tid < L::T; ++tid) L::phaseC(g, tile, tid, lds.data(), L::load_uniform(g));
            }
        }, cs, lds_rows);
        CHECK(okl, "lds transform rejected log_n=%d", plan.log_n);
        return;
    }
    bool ok = for_each_pass<LQ>(plan, blob.data(), inverse, src, work, dst, batch, [&](auto pass, auto lzc, const PassArgs& a, uint64_t nblocks) {
        using P = decltype(pass);
        constexpr int LZ = decltype(lzc)::value;
        std::vector<uint32_t> lds(P::LDS_WORDS, 0xDEADBEEFu);  // exact size: an out-of-range LDS word is an ASan error
        std::vector<char> seen(nblocks, 0);
        for (uint64_t v = 0; v < nblocks; ++v) {
            const uint32_t b = P::tile_order((uint32_t)v, (uint32_t)nblocks);  // the persistent loop's virtual index -> tile
            CHECK(b < nblocks && !seen[b], "tile_order is not a bijection: v=%llu -> %u of %llu", (unsigned long long)v, b, (unsigned long long)nblocks);
            if (b < nblocks) seen[b] = 1;
            emu_tile<P, LZ>(a, b, lds.data());
        }
    }, cs, lde_log);
    CHECK(ok, "no pass instantiation for log_n=%d", plan.log_n);
}

// BabyBearDomain::fft / ifft on a coset (src/math/domain.rs:85-123) with the scaling fused into the passes
static void test_coset(int log_n, uint64_t batch, uint32_t shift) {
    NttPlan plan;
    CHECK(build_plan(log_n, plan, latency_plan), "plan %d", log_n);
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> ref(n * batch), want(n * batch);
    orc_fill_splitmix(ref.data(), n * batch, 0xC05E7ull + log_n);
    std::vector<uint32_t> in(n * batch), work(n * batch), out(n * batch);
    for (size_t i = 0; i < n * batch; ++i) in[i] = (uint32_t)ref[i];
    for (uint64_t b = 0; b < batch; ++b) orc_domain_fft(want.data() + b * n, n, ref.data() + b * n, n, shift);
    emu_transform(plan, false, in.data(), work.data(), out.data(), batch, shift);
    size_t bad = 0;
    for (size_t i = 0; i < n * batch; ++i) if (out[i] != (uint32_t)want[i]) ++bad;
    CHECK(bad == 0, "coset fft log_n=%d batch=%llu: %zu mismatches", log_n, (unsigned long long)batch, bad);
    // ifft of the same input (as evaluations), in place
    want = ref;
    for (uint64_t b = 0; b < batch; ++b) orc_domain_ifft(want.data() + b * n, n, shift);
    std::vector<uint32_t> buf = in;
    emu_transform(plan, true, buf.data(), work.data(), buf.data(), batch, shift);
    bad = 0;
    for (size_t i = 0; i < n * batch; ++i) if (buf[i] != (uint32_t)want[i]) ++bad;
    CHECK(bad == 0, "coset ifft log_n=%d batch=%llu: %zu mismatches", log_n, (unsigned long long)batch, bad);
}

// fft_ext / ifft_ext (src/math/domain.rs:129-151): `vectors` Ext vectors of n elements in the reference's AoS layout ([n][4]); the
// interleaved passes must give, coordinate by coordinate, the oracle's base transforms.  lde_log > 0: the forward transform of a
// compact coefficient vector ([n >> lde_log][4]) with the padding implied.
static void test_ext(int log_n, uint64_t vectors, uint32_t shift, int lde_log = 0) {
    NttPlan plan;
    CHECK(build_plan(log_n, plan, latency_plan && (log_n == 21 || (log_n == 22 && lde_log))), "plan %d", log_n);   // "Q1": 2^21, and the LDE to 2^22, through their two-pass plans
    const size_t n = (size_t)1 << log_n, n_in = n >> lde_log;
    std::vector<uint64_t> ref(4 * n_in * vectors);
    orc_fill_splitmix(ref.data(), ref.size(), 0xE7700ull + (uint64_t)log_n * 131 + (uint64_t)lde_log);
    std::vector<uint32_t> in(ref.size()), work(4 * n * vectors, 0xABABABABu), out(4 * n * vectors, 0xCDCDCDCDu);
    for (size_t i = 0; i < ref.size(); ++i) in[i] = (uint32_t)ref[i];
    std::vector<uint64_t> col(n_in), want(n);
    // forward (coset shift fused, padding implied), out of place
    emu_transform<2>(plan, false, in.data(), work.data(), out.data(), 4 * vectors, shift, lde_log);
    size_t bad = 0;
    for (uint64_t v = 0; v < vectors; ++v)
        for (int k = 0; k < 4; ++k) {
            for (size_t j = 0; j < n_in; ++j) col[j] = ref[(v * n_in + j) * 4 + k];
            orc_domain_fft(want.data(), n, col.data(), n_in, shift);
            for (size_t j = 0; j < n; ++j) bad += out[(v * n + j) * 4 + k] != (uint32_t)want[j];
        }
    CHECK(bad == 0, "ext forward log_n=%d vectors=%llu shift=%u lde_log=%d: %zu mismatches", log_n, (unsigned long long)vectors, shift, lde_log, bad);
    if (lde_log) return;
    // inverse of the same data (as evaluations), in place
    std::vector<uint32_t> buf = in;
    emu_transform<2>(plan, true, buf.data(), work.data(), buf.data(), 4 * vectors, shift);
    bad = 0;
    for (uint64_t v = 0; v < vectors; ++v)
        for (int k = 0; k < 4; ++k) {
            for (size_t j = 0; j < n; ++j) want[j] = ref[(v * n + j) * 4 + k];
            orc_domain_ifft(want.data(), n, shift);
            for (size_t j = 0; j < n; ++j) bad += buf[(v * n + j) * 4 + k] != (uint32_t)want[j];
        }
    CHECK(bad == 0, "ext inverse log_n=%d vectors=%llu shift=%u: %zu mismatches", log_n, (unsigned long long)vectors, shift, bad);
}

static void test_field() {
    uint64_t s = 12345;
    auto next = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)((s >> 20) % BB_P); };
    std::vector<uint32_t> edge = {0u, 1u, 2u, BB_P - 1, BB_P - 2, BB_HALF, BB_R1, BB_R2, 0x7FFFFFFFu % BB_P};
    for (int it = 0; it < 200000; ++it) {
        uint32_t a = it < 81 ? edge[it / 9] : next();
        uint32_t b = it < 81 ? edge[it % 9] : next();
        uint32_t want = (uint32_t)orc_bb_mul(a, b);
        CHECK(mont_mul(a, to_mont(b)) == want, "mont_mul %u %u", a, b);
        CHECK(mont_mul(a, to_mont_host(b)) == want, "to_mont_host %u %u", a, b);
        CHECK(bb_mul_barrett64(a, b) == want, "barrett %u %u", a, b);
        CHECK(bb_mul_plain(a, b) == want, "mul_plain %u %u", a, b);
        CHECK(bb_add(a, b) == (uint32_t)orc_bb_add(a, b), "add %u %u", a, b);
        CHECK(bb_sub(a, b) == (uint32_t)orc_bb_sub(a, b), "sub %u %u", a, b);
        CHECK(mont_mul(bb_sub_lazy(a, b), to_mont(b)) == (uint32_t)orc_bb_mul(orc_bb_sub(a, b), b), "sub_lazy %u %u", a, b);
        CHECK(bb_halve(a) == (uint32_t)orc_bb_mul(a, BB_HALF), "halve %u", a);
        if (b) CHECK(mont_dot_sub(a, (uint32_t)((a * 2654435761ull) % BB_P), to_mont(b), BB_P - to_mont(b)) ==
                     (uint32_t)orc_bb_mul(orc_bb_sub(a, (a * 2654435761ull) % BB_P), b), "dot_sub %u %u", a, b);
        // lazy product accepts any u32 on the left
        uint32_t any = a * 2654435761u;
        CHECK(bb_reduce_2p(mont_mul_lazy(any, to_mont(b))) == (uint32_t)orc_bb_mul(any % BB_P, b), "lazy any %u %u", any, b);
        if (it < 2000 && a) CHECK(from_mont(mont_inv_chain(to_mont(a))) == (uint32_t)orc_bb_inverse(a), "inv %u", a);
    }
    CHECK(from_mont(BB_R1) == 1u, "R1");
    CHECK(narrow_u64((uint64_t)BB_P + 5) == 5u, "narrow");
}

static void test_ntt(int log_n, uint64_t batch, int pattern) {
    NttPlan plan;
    CHECK(build_plan(log_n, plan, latency_plan), "plan %d", log_n);
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> ref(n * batch);
    if (pattern == 0) orc_fill_splitmix(ref.data(), n * batch, 0x70796E69ull + ((uint64_t)log_n << 32));
    else for (uint64_t b = 0; b < batch; ++b) orc_fill_pattern_7i3(ref.data() + b * n, n);
    std::vector<uint32_t> in(n * batch), work(n * batch, 0xABABABABu), out(n * batch, 0xCDCDCDCDu);
    for (size_t i = 0; i < n * batch; ++i) in[i] = (uint32_t)ref[i];

    // forward, out of place
    emu_transform(plan, false, in.data(), work.data(), out.data(), batch);
    std::vector<uint64_t> fwd = ref;
    for (uint64_t b = 0; b < batch; ++b) orc_ntt_canonical(fwd.data() + b * n, n);
    size_t bad = 0;
    for (size_t i = 0; i < n * batch; ++i) if (out[i] != (uint32_t)fwd[i]) { if (!bad) std::printf("  first fwd mismatch log_n=%d i=%zu got=%u want=%u\n", log_n, i, out[i], (uint32_t)fwd[i]); ++bad; }
    CHECK(bad == 0, "forward log_n=%d batch=%llu: %zu mismatches", log_n, (unsigned long long)batch, bad);

    // inverse of `in` (as evaluations), in place
    std::vector<uint32_t> buf = in;
    emu_transform(plan, true, buf.data(), work.data(), buf.data(), batch);
    std::vector<uint64_t> inv = ref;
    for (uint64_t b = 0; b < batch; ++b) orc_intt_canonical(inv.data() + b * n, n);
    bad = 0;
    for (size_t i = 0; i < n * batch; ++i) if (buf[i] != (uint32_t)inv[i]) { if (!bad) std::printf("  first inv mismatch log_n=%d i=%zu got=%u want=%u\n", log_n, i, buf[i], (uint32_t)inv[i]); ++bad; }
    CHECK(bad == 0, "inverse log_n=%d batch=%llu: %zu mismatches", log_n, (unsigned long long)batch, bad);

    // roundtrip in place: intt(ntt(x)) == x  (src/ntt.rs:289-310)
    emu_transform(plan, false, out.data(), work.data(), out.data(), batch);   // out = ntt(ntt(x)) -- just exercising in-place forward
    buf = in;
    emu_transform(plan, false, buf.data(), work.data(), buf.data(), batch);
    emu_transform(plan, true, buf.data(), work.data(), buf.data(), batch);
    CHECK(buf == in, "roundtrip log_n=%d", log_n);
}

static void test_fold(int log_N, int layer, uint32_t shift) {
    NttPlan plan;
    CHECK(build_plan(log_N, plan), "plan");
    const size_t N = (size_t)1 << log_N, m = N >> layer, half = m / 2;
    std::vector<uint64_t> evals(m), xs(N), want(half);
    orc_fill_splitmix(evals.data(), m, 77 + layer);
    orc_domain_elements(xs.data(), N, shift);
    for (int k = 0; k < layer; ++k) for (size_t i = 0; i < N; ++i) xs[i] = orc_bb_mul(xs[i], xs[i]);  // src/fibonacci.rs:228-231
    const uint64_t beta = 1234567 + layer;
    orc_fri_fold(want.data(), evals.data(), m, xs.data(), beta);

    std::vector<uint32_t> e32(m), out(half);
    for (size_t i = 0; i < m; ++i) e32[i] = (uint32_t)evals[i];
    FoldArgs f{};
    f.evals = e32.data();
    f.out = out.data();
    f.dom = sub_domain(plan, plan.inv.data(), layer);
    const uint32_t x0 = (uint32_t)xs[0];
    f.coef = to_mont_host(bb_mul_host(bb_mul_host((uint32_t)beta, BB_HALF), bb_inv_host(x0)));
    f.half = half;
    f.step = to_mont_host(bb_inv_host(bb_root_of_unity_host((uint32_t)(log_N - layer))));
    for (size_t i = 0; i < half; ++i) out[i] = fold_one(f, i, e32[i], e32[i + half]);
    size_t bad = 0;
    for (size_t i = 0; i < half; ++i) if (out[i] != (uint32_t)want[i]) ++bad;
    CHECK(bad == 0, "fold log_N=%d layer=%d: %zu mismatches", log_N, layer, bad);
    for (size_t i = 0; i + 4 <= half; i += 4) {   // the four-outputs form of the large-layer stream: one lookup, three running products
        const uint32_t a4[4] = {e32[i], e32[i + 1], e32[i + 2], e32[i + 3]}, b4[4] = {e32[i + half], e32[i + half + 1], e32[i + half + 2], e32[i + half + 3]};
        uint32_t r4[4];
        fold_quad(f, i, a4, b4, r4);
        for (int j = 0; j < 4; ++j) if (r4[j] != (uint32_t)want[i + j]) ++bad;
    }
    CHECK(bad == 0, "fold_quad log_N=%d layer=%d: %zu mismatches", log_N, layer, bad);
}

// fold_xs_batch (B outputs behind one inversion; points plain, the shared inverse carries beta / 2) against the oracle's fri_fold on
// explicit points, with zero points inside a batch (their own inverse is 0; their neighbours' must be untouched); and
// batch_inverse_scaled itself against the oracle's inverse, incl. the extreme residues
template <int B>
static void fold_xs_batch_case(uint64_t seed) {
    const size_t m = 64, half = m / 2;
    std::vector<uint64_t> evals(m), xs(m), want(half);
    orc_fill_splitmix(evals.data(), m, seed);
    orc_fill_splitmix(xs.data(), m, seed + 101);
    for (size_t i = 0; i < m; ++i) if (xs[i] == 0) xs[i] = 1;
    xs[0] = 1; xs[1] = BB_P - 1; xs[2] = 2; evals[0] = 0; evals[half] = BB_P - 1;    // extreme operands
    const uint64_t beta = seed % 3 == 0 ? 0 : (seed % 3 == 1 ? BB_P - 1 : 987654321);
    orc_fri_fold(want.data(), evals.data(), m, xs.data(), beta);
    const uint32_t beta_half = bb_mul_host((uint32_t)beta, BB_HALF);
    size_t bad = 0;
    for (size_t g = 0; g < half / B; ++g) {
        uint32_t x[B], a[B], b[B], r[B];
        for (int j = 0; j < B; ++j) { x[j] = (uint32_t)xs[B * g + j]; a[j] = (uint32_t)evals[B * g + j]; b[j] = (uint32_t)evals[B * g + j + half]; }
        fold_xs_batch<B>(x, a, b, beta_half, r);
        for (int j = 0; j < B; ++j) bad += r[j] != (uint32_t)want[B * g + j];
        uint32_t inv[B];
        batch_inverse_scaled<B>(x, 1u, inv);                                          // x^-1 R
        for (int j = 0; j < B; ++j) bad += from_mont(inv[j]) != (uint32_t)orc_bb_inverse(x[j]);
        // zero points at the first, an inner and the last position: their outputs are the plain average, everyone else's are unchanged
        const int z0 = 0, z1 = B > 2 ? B / 2 : 0, z2 = B - 1;
        x[z0] = 0; x[z1] = 0; x[z2] = 0;
        uint32_t rz[B];
        fold_xs_batch<B>(x, a, b, beta_half, rz);
        for (int j = 0; j < B; ++j) {
            const uint32_t avg = bb_halve(bb_add(a[j], b[j]));
            bad += (j == z0 || j == z1 || j == z2) ? rz[j] != avg : rz[j] != r[j];
        }
    }
    CHECK(bad == 0, "fold_xs_batch<%d> seed %llu: %zu mismatches", B, (unsigned long long)seed, bad);
}
static void test_fold_xs_batch() {
    for (uint64_t seed = 4242; seed < 4242 + 12; ++seed) {
        fold_xs_batch_case<16>(seed);
        fold_xs_batch_case<4>(seed);
        fold_xs_batch_case<1>(seed);
    }
}

static void test_fold_ext(size_t len) {
    const size_t half = len / 2;
    std::vector<uint64_t> evals(4 * len), xs(len), want(4 * half);
    orc_fill_splitmix(evals.data(), 4 * len, 4242);
    orc_domain_elements(xs.data(), len, 7);
    const uint64_t beta[4] = {123456789ull, 987654321ull, 5ull, BB_P - 1ull};
    orc_fri_fold_ext(want.data(), evals.data(), len, xs.data(), beta);
    uint32_t bh[4];
    for (int k = 0; k < 4; ++k) bh[k] = bb_mul_host((uint32_t)beta[k], BB_HALF);
    const ExtFactor f = ext_factor_host(bh);
    size_t bad = 0;
    for (size_t i = 0; i < half; ++i) {
        Ext4 a, b;
        for (int k = 0; k < 4; ++k) { a.c[k] = (uint32_t)evals[4 * i + k]; b.c[k] = (uint32_t)evals[4 * (i + half) + k]; }
        const Ext4 r = fold_ext_one(a, b, to_mont_host(bb_inv_host((uint32_t)xs[i])), f);
        for (int k = 0; k < 4; ++k) if (r.c[k] != (uint32_t)want[4 * i + k]) ++bad;
    }
    CHECK(bad == 0, "fold_ext len=%zu: %zu mismatches", len, bad);
}

// Merkle: leaf / node digests of merkle_kernels.hpp vs the oracle's byte-level tree (src/merkle.rs:25-48,105-123)
static void test_merkle(size_t n, bool salted) {
    std::vector<uint64_t> vals(n);
    orc_fill_splitmix(vals.data(), n, 0x3E2C1Eull + n);
    std::vector<uint8_t> salts(16 * n);
    for (size_t i = 0; i < salts.size(); ++i) salts[i] = (uint8_t)(i * 131 + 7);
    const size_t total = orc_merkle_total_digests(n);
    std::vector<uint8_t> want(32 * total);
    orc_merkle_commit_values(want.data(), vals.data(), salted ? salts.data() : nullptr, n);
    std::vector<Digest> lv(total);
    for (size_t i = 0; i < n; ++i) {
        uint32_t sw[4];
        std::memcpy(sw, salts.data() + 16 * i, 16);
        lv[i] = merkle_leaf((uint32_t)vals[i], salted ? sw : nullptr);
    }
    size_t off = 0, m = n;
    while (m > 1) {
        const size_t up = (m + 1) / 2;
        for (size_t i = 0; i < up; ++i) lv[off + m + i] = merkle_node(lv[off + 2 * i], lv[off + (2 * i + 1 < m ? 2 * i + 1 : 2 * i)]);
        off += m;
        m = up;
    }
    CHECK(std::memcmp(lv.data(), want.data(), 32 * total) == 0, "merkle n=%zu salted=%d", n, (int)salted);
}

// The two-wave node hash (merkle_kernels.hpp, "a node hash on TWO waves"): the phase bodies run in the kernel's order -- helper A,
// main A, (barrier), helper B, main B, (barrier), main C -- for 64 lanes sharing one hand-over area, against merkle_node; and the
// compile-time table of block 2's schedule against a direct expansion.
static void test_merkle_coop() {
    static const ShaB2Table b2 = make_sha_b2_table();
    std::vector<uint32_t> sched(COOP_SCHED_WORDS, 0xDEADBEEFu);
    Digest l[64], r[64];
    uint64_t seed = 0xC00Full;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull; l[lane].m[j] = (uint32_t)(seed >> 32);
            seed = seed * 6364136223846793005ull + 1442695040888963407ull; r[lane].m[j] = (uint32_t)(seed >> 32);
        }
    r[7] = l[7];                                     // an odd level's duplicated last node
    for (int j = 0; j < 8; ++j) r[9].m[j] = 0xFFFFFFFFu;
    uint32_t wh[64][16], wm[64][16], last[64];
    ShaRegs st[64];
    for (uint32_t lane = 0; lane < 64; ++lane) {     // phase A
        uint32_t dummy;
        node_block1(l[lane], r[lane], wh[lane], dummy);
        coop_helper_schedule<16, 24>(wh[lane], sched.data(), lane);
        node_block1(l[lane], r[lane], wm[lane], last[lane]);
        coop_main_first16(st[lane], wm[lane]);
    }
    for (uint32_t lane = 0; lane < 64; ++lane) {     // phase B
        coop_helper_schedule<40, 24>(wh[lane], sched.data(), lane);
        coop_main_rounds<16>(st[lane], sched.data(), lane);
    }
    size_t bad = 0;
    for (uint32_t lane = 0; lane < 64; ++lane) {     // phase C
        coop_main_rounds<40>(st[lane], sched.data(), lane);
        uint32_t kw2[64];
        for (int i = 0; i < 64; ++i) kw2[i] = b2.kw[i][last[lane]];
        const Digest got = coop_main_finish(st[lane], kw2);
        const Digest want = merkle_node(l[lane], r[lane]);
        bad += std::memcmp(&got, &want, sizeof(Digest)) != 0;
    }
    CHECK(bad == 0, "two-wave node hash: %zu of 64 digests differ from merkle_node", bad);
    for (size_t k = 0; k < COOP_SCHED_WORDS; ++k) bad += sched[k] == 0xDEADBEEFu;
    CHECK(bad == 0, "two-wave node hash: %zu hand-over words never written", bad);
}

// Low-degree extension: compact input (n >> lde_log words per transform), zero padding implied, coset shift fused
static void test_lde(int log_n, uint64_t batch, int lde_log, uint32_t shift) {
    NttPlan plan;
    CHECK(build_plan(log_n, plan, latency_plan), "plan %d", log_n);
    const uint64_t n = 1ull << log_n, n_in = n >> lde_log;
    std::vector<uint64_t> c64(n_in * batch);
    orc_fill_splitmix(c64.data(), c64.size(), 0x1DE0000ull + (uint64_t)log_n * 64 + (uint64_t)lde_log);
    std::vector<uint32_t> in(n_in * batch), out(n * batch, 0xDEADBEEFu), work(n * batch);  // `in` is EXACTLY the compact size
    for (size_t i = 0; i < in.size(); ++i) in[i] = (uint32_t)c64[i];
    emu_transform(plan, false, in.data(), work.data(), out.data(), batch, shift, lde_log);
    uint64_t bad = 0;
    std::vector<uint64_t> want(n);
    for (uint64_t b = 0; b < batch; ++b) {
        orc_domain_fft(want.data(), n, c64.data() + b * n_in, n_in, shift);
        for (uint64_t k = 0; k < n; ++k) bad += out[b * n + k] != (uint32_t)want[k];
    }
    CHECK(bad == 0, "lde log_n=%d batch=%llu lde_log=%d shift=%u: %llu mismatches", log_n, (unsigned long long)batch, lde_log, shift,
          (unsigned long long)bad);
}

// One transform over G ranks (include/toyni_hip.h 2b), all ranks stepped in this process: slab pass on every rank's
// column block, the all-to-all as memcpy, relayout, size-S1 row transforms -- and the mirrored inverse.
static void emu_slab_pass(const NttPlan& plan, bool inverse, uint32_t* slab, uint64_t cols, uint64_t col_base, const uint32_t* ones) {
    const std::vector<uint32_t>& blob = inverse ? plan.inv : plan.fwd;
    bool ok = slab_pass(plan, blob.data(), inverse, slab, cols, col_base, ones, [&](auto pass, const PassArgs& a, uint64_t nblocks) {
        using P = decltype(pass);
        std::vector<uint32_t> lds(P::LDS_WORDS, 0xDEADBEEFu);
        for (uint64_t v = 0; v < nblocks; ++v) emu_tile<P>(a, P::tile_order((uint32_t)v, (uint32_t)nblocks), lds.data());
    });
    CHECK(ok, "slab pass rejected: log_n=%d cols=%llu", plan.log_n, (unsigned long long)cols);
}
static void emu_relayout(const NttPlan& plan, const uint32_t* in, uint32_t* out, uint64_t rows, uint64_t row0, uint64_t parts, bool inverse) {
    const uint64_t s1 = (1ull << plan.log_n) >> plan.pass[0].log_m;
    auto lg = [](uint64_t v) { int l = 0; while ((1ull << l) < v) ++l; return (uint32_t)l; };
    RelayoutArgs a{};
    a.in = in; a.out = out;
    a.log_rows = lg(rows); a.log_parts = lg(parts); a.log_w = lg(s1 / parts);
    a.inverse = inverse; a.row0 = (uint32_t)row0;
    a.lo = plan.inv.data() + plan.dom_lo_off; a.hi = plan.inv.data() + plan.dom_hi_off; a.lowbits = plan.dom_lowbits;
    for (uint64_t o = 0; o < rows * s1; ++o) {
        uint32_t e;
        const uint64_t src = relayout_src(a, o, e);
        out[o] = relayout_value(a, in[src], e);
    }
}
// toyni_ntt_slab_rows_device: the row transforms addressing the pieces layout directly (PassArgs::in_split_* / out_split_*, cs_mode 4).
// Returns false when the dispatched shapes cannot (the library then runs relayout + transform; test_slab checks that form anyway).
static unsigned long long slab_rows_fused = 0, slab_rows_fallback = 0;
static bool emu_slab_rows(const NttPlan& big, const NttPlan& sub, bool inverse, const uint32_t* in, uint32_t* work, uint32_t* out, uint64_t rows,
                          uint64_t row0, uint64_t parts) {
    auto lg = [](uint64_t v) { int l = 0; while ((1ull << l) < v) ++l; return (uint32_t)l; };
    SlabIo sio;
    bool unsupported = false;
    sio.log_parts = lg(parts);
    sio.tw_lo = big.inv.data() + big.dom_lo_off;
    sio.tw_hi = big.inv.data() + big.dom_hi_off;
    sio.tw_lowbits = big.dom_lowbits;
    sio.row0 = (uint32_t)row0;
    sio.unsupported = &unsupported;
    sio.dry = true;
    const std::vector<uint32_t>& blob = inverse ? sub.inv : sub.fwd;
    bool ok = sub.npasses >= 2 && rows >= 2 &&
              for_each_pass<0>(sub, blob.data(), inverse, in, work, out, rows, [](auto, auto, const PassArgs&, uint64_t) {}, CosetTables(), 0, false, &sio);
    if (!ok || unsupported) { ++slab_rows_fallback; return false; }
    sio.dry = false;
    ok = for_each_pass<0>(sub, blob.data(), inverse, in, work, out, rows, [&](auto pass, auto lzc, const PassArgs& a, uint64_t nblocks) {
        using P = decltype(pass);
        constexpr int LZ = decltype(lzc)::value;
        std::vector<uint32_t> lds(P::LDS_WORDS, 0xDEADBEEFu);
        for (uint64_t v = 0; v < nblocks; ++v) emu_tile<P, LZ>(a, P::tile_order((uint32_t)v, (uint32_t)nblocks), lds.data());
    }, CosetTables(), 0, false, &sio);
    CHECK(ok && !unsupported, "slab rows: dry run said yes, the run said no");
    ++slab_rows_fused;
    return true;
}

static void test_slab(int log_n, uint64_t G) {
    NttPlan plan, sub;
    CHECK(build_plan(log_n, plan), "plan %d", log_n);
    const uint64_t n = 1ull << log_n, m1 = 1ull << plan.pass[0].log_m, s1 = n / m1;
    CHECK(build_plan(log_n - plan.pass[0].log_m, sub), "sub plan");
    const uint64_t W = s1 / G, R = m1 / G;
    std::vector<uint64_t> x(n), want(n);
    orc_fill_splitmix(x.data(), n, 0x51AB0000ull + (uint64_t)log_n * 16 + G);
    want = x;
    orc_ntt_canonical(want.data(), n);
    const uint32_t lb = plan.pass[0].lowbits;  // exactly the size the library allocates: an overrun is an ASan error
    std::vector<uint32_t> ones((size_t)1 << (lb > (uint32_t)log_n - lb ? lb : (uint32_t)log_n - lb), to_mont_host(1u));
    // rank g's slab [M1][W]
    std::vector<std::vector<uint32_t>> slab(G, std::vector<uint32_t>(m1 * W)), recv(G, std::vector<uint32_t>(m1 * W)), rows(G, std::vector<uint32_t>(R * s1));
    for (uint64_t g = 0; g < G; ++g)
        for (uint64_t j1 = 0; j1 < m1; ++j1)
            for (uint64_t c = 0; c < W; ++c) slab[g][j1 * W + c] = (uint32_t)x[j1 * s1 + g * W + c];
    std::vector<std::vector<uint32_t>> slab0 = slab;
    for (uint64_t g = 0; g < G; ++g) emu_slab_pass(plan, false, slab[g].data(), W, g * W, ones.data());
    // all_to_all_single: rank g's send block h (rows k1 in rank h's block, contiguous) -> rank h's recv block g
    for (uint64_t g = 0; g < G; ++g)
        for (uint64_t h = 0; h < G; ++h) std::memcpy(&recv[h][g * R * W], &slab[g][h * R * W], R * W * sizeof(uint32_t));
    std::vector<uint32_t> work(R * s1);
    for (uint64_t g = 0; g < G; ++g) {
        emu_relayout(plan, recv[g].data(), rows[g].data(), R, g * R, G, false);
        emu_transform(sub, false, rows[g].data(), work.data(), rows[g].data(), R);
        {   // the fused form of the same two steps must produce the same rows (exactly sized buffers: an overrun is an ASan error)
            std::vector<uint32_t> fused(R * s1, 0xFEFEFEFEu), fwork(R * s1);
            if (emu_slab_rows(plan, sub, false, recv[g].data(), fwork.data(), fused.data(), R, g * R, G))
                CHECK(fused == rows[g], "slab rows (fused, forward) log_n=%d G=%llu rank=%llu", log_n, (unsigned long long)G, (unsigned long long)g);
        }
        uint64_t bad = 0;
        for (uint64_t r = 0; r < R; ++r)
            for (uint64_t k = 0; k < s1; ++k) bad += rows[g][r * s1 + k] != (uint32_t)want[(g * R + r) + m1 * k];
        CHECK(bad == 0, "slab forward log_n=%d G=%llu rank=%llu: %llu mismatches", log_n, (unsigned long long)G, (unsigned long long)g, (unsigned long long)bad);
    }
    // inverse: back to the input slabs
    std::vector<std::vector<uint32_t>> send(G, std::vector<uint32_t>(R * s1));
    for (uint64_t g = 0; g < G; ++g) {
        std::vector<uint32_t> fused(R * s1, 0xFEFEFEFEu), fwork(R * s1);
        const bool did = emu_slab_rows(plan, sub, true, rows[g].data(), fwork.data(), fused.data(), R, g * R, G);   // rows[g] is only read
        emu_transform(sub, true, rows[g].data(), work.data(), rows[g].data(), R);
        emu_relayout(plan, rows[g].data(), send[g].data(), R, g * R, G, true);
        if (did) CHECK(fused == send[g], "slab rows (fused, inverse) log_n=%d G=%llu rank=%llu", log_n, (unsigned long long)G, (unsigned long long)g);
    }
    for (uint64_t g = 0; g < G; ++g)
        for (uint64_t h = 0; h < G; ++h) std::memcpy(&slab[h][g * R * W], &send[g][h * R * W], R * W * sizeof(uint32_t));
    for (uint64_t g = 0; g < G; ++g) {
        emu_slab_pass(plan, true, slab[g].data(), W, g * W, ones.data());
        CHECK(slab[g] == slab0[g], "slab inverse log_n=%d G=%llu rank=%llu", log_n, (unsigned long long)G, (unsigned long long)g);
    }
}

// Pointwise prover steps (prover_kernels.hpp) against the oracle's restatement of src/fibonacci.rs:133-150,186-198
static DomainArgs emu_domain(const NttPlan& plan, int log_m, uint32_t shift) {
    DomainArgs d{};
    d.dom = sub_domain(plan, plan.fwd.data(), plan.log_n - log_m);
    d.shiftR = to_mont_host(shift);
    return d;
}
static void test_prover_steps(int log_N, int log_blowup, uint32_t shift) {
    NttPlan plan;
    CHECK(build_plan(log_N, plan), "plan");
    const size_t N = (size_t)1 << log_N, n = N >> log_blowup, B = (size_t)1 << log_blowup;
    std::vector<uint64_t> lde(N), cw(N), qw(N), dw(N);
    orc_fill_splitmix(lde.data(), N, 0xF1B0 + log_N);
    CHECK(orc_fib_quotient(cw.data(), qw.data(), lde.data(), N, n, shift) == 0, "oracle quotient");
    std::vector<uint32_t> t32(N);
    for (size_t i = 0; i < N; ++i) t32[i] = (uint32_t)lde[i];
    QuotientArgs a{};
    a.trace = t32.data();
    a.dom = emu_domain(plan, log_N, shift);
    a.log_N = (uint32_t)log_N;
    a.log_blowup = (uint32_t)log_blowup;
    const uint32_t g = (uint32_t)orc_bb_root_of_unity((uint32_t)(log_N - log_blowup));
    a.b1R = to_mont_host(bb_pow_host(g, n - 1));
    a.b2R = to_mont_host(bb_pow_host(g, n - 2));
    a.shift_nR = to_mont_host(bb_pow_host(shift, n));
    a.wBR = to_mont_host(bb_pow_host((uint32_t)orc_bb_root_of_unity((uint32_t)log_N), n));
    size_t bad = 0;
    std::vector<uint32_t> q32(N);
    for (size_t i = 0; i < N; ++i) {
        uint32_t c, q;
        quotient_one(a, i, t32[i], t32[(i + B) % N], t32[(i + 2 * B) % N], quotient_zh_inv(a, (uint32_t)(i & (B - 1))), c, q);
        q32[i] = q;
        bad += c != (uint32_t)cw[i] || q != (uint32_t)qw[i];
    }
    CHECK(bad == 0, "quotient log_N=%d blowup=%d: %zu mismatches", log_N, log_blowup, bad);
    const uint64_t z = 7654321, tz = 11, tgz = 22, tggz = 33, qz = 44;
    CHECK(orc_fib_deep(dw.data(), lde.data(), qw.data(), N, n, shift, z, tz, tgz, tggz, qz) == 0, "oracle deep");
    DeepArgs d{};
    d.trace = t32.data();
    d.quot = q32.data();
    d.dom = a.dom;
    d.log_N = (uint32_t)log_N;
    d.log_blowup = (uint32_t)log_blowup;
    d.wNR = to_mont_host((uint32_t)orc_bb_root_of_unity((uint32_t)log_N));
    d.zR = to_mont_host((uint32_t)z);
    d.t_z = tz; d.t_gz = tgz; d.t_ggz = tggz; d.q_z = qz;
    bad = 0;
    for (size_t i0 = 0; i0 + 8 <= N; i0 += 8) {
        uint32_t t0[8], t1[8], t2[8], qv[8], out[8];
        for (int j = 0; j < 8; ++j) { t0[j] = t32[i0 + j]; t1[j] = t32[(i0 + j + B) % N]; t2[j] = t32[(i0 + j + 2 * B) % N]; qv[j] = q32[i0 + j]; }
        deep_group<8>(d, i0, t0, t1, t2, qv, out);
        for (int j = 0; j < 8; ++j) bad += out[j] != (uint32_t)dw[i0 + j];
    }
    CHECK(bad == 0, "deep log_N=%d blowup=%d: %zu mismatches", log_N, log_blowup, bad);
    // a point with x_i = z: that point alone yields 0, its seven neighbours are unaffected
    {
        std::vector<uint64_t> xs(N);
        orc_domain_elements(xs.data(), N, shift);
        d.zR = to_mont_host((uint32_t)xs[3]);
        uint32_t t0[8], t1[8], t2[8], qv[8], out[8];
        for (int j = 0; j < 8; ++j) { t0[j] = t32[j]; t1[j] = t32[(j + B) % N]; t2[j] = t32[(j + 2 * B) % N]; qv[j] = q32[j]; }
        deep_group<8>(d, 0, t0, t1, t2, qv, out);
        CHECK(out[3] == 0u, "deep zero point");
        for (int j = 0; j < 8; ++j) {
            if (j == 3) continue;
            const uint64_t num = orc_bb_add(orc_bb_add(orc_bb_sub(qv[j], qz), orc_bb_sub(t2[j], tggz)), orc_bb_add(orc_bb_sub(t1[j], tgz), orc_bb_sub(t0[j], tz)));
            CHECK(out[j] == (uint32_t)orc_bb_mul(num, orc_bb_inverse(orc_bb_sub(xs[j], xs[3]))), "deep neighbour of a zero point j=%d", j);
        }
    }
}
static void test_poly_eval(size_t ncoeffs, uint32_t z) {
    std::vector<uint64_t> c64(ncoeffs);
    orc_fill_splitmix(c64.data(), ncoeffs, 0x9017 + ncoeffs);
    const uint64_t want = orc_poly_eval(c64.data(), ncoeffs, z);
    PolyEvalArgs a{};
    a.npoints = 1;
    a.zR[0] = to_mont_host(z);
    a.z16R[0] = to_mont_host(bb_pow_host(z, POLY_PER_THREAD));
    a.zchunkR[0] = to_mont_host(bb_pow_host(z, POLY_CHUNK));
    const size_t nblocks = (ncoeffs + POLY_CHUNK - 1) / POLY_CHUNK;
    uint32_t total = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        uint32_t part = 0;
        for (uint32_t t = 0; t < POLY_THREADS; ++t) {
            uint32_t c[POLY_PER_THREAD];
            for (uint32_t j = 0; j < POLY_PER_THREAD; ++j) {
                const size_t i = b * POLY_CHUNK + (size_t)t * POLY_PER_THREAD + j;
                c[j] = i < ncoeffs ? (uint32_t)c64[i] : 0u;
            }
            part = bb_add(part, poly_thread_term(a, 0, c, t));
        }
        total = bb_add(total, mont_mul(part, mont_pow(a.zchunkR[0], b)));
    }
    CHECK(total == (uint32_t)want, "poly eval ncoeffs=%zu z=%u: got %u want %u", ncoeffs, z, total, (uint32_t)want);
}
static void test_merkle_open(size_t n) {
    std::vector<uint64_t> vals(n);
    orc_fill_splitmix(vals.data(), n, 0x0BE7 + n);
    const size_t total = orc_merkle_total_digests(n);
    std::vector<uint8_t> levels(32 * total), path(32 * 64), pos(64);
    orc_merkle_commit_values(levels.data(), vals.data(), nullptr, n);
    const uint32_t depth = merkle_depth(n);
    for (size_t index = 0; index < n; ++index) {
        const int d = orc_merkle_get_proof(path.data(), pos.data(), levels.data(), n, index);
        CHECK(d == (int)depth, "depth n=%zu", n);
        for (uint32_t l = 0; l < depth; ++l) {
            bool is_left;
            const uint64_t row = merkle_sibling_row(n, index, l, is_left);
            CHECK(row < total && std::memcmp(levels.data() + 32 * row, path.data() + 32 * l, 32) == 0 && is_left == (pos[l] != 0),
                  "merkle open n=%zu index=%zu level=%u", n, index, l);
        }
    }
    CHECK(merkle_open_record_bytes(n) % 8 == 0, "record size");
}

int main(int argc, char** argv) {
    int max_log = argc > 1 ? std::atoi(argv[1]) : 16;
    test_field();
    std::printf("field ok=%d\n", failures == 0);
    for (int log_n = 0; log_n <= max_log; ++log_n) {
        if (log_n == 11 || log_n == 12) {           // n = 2^11 / 2^12: the plans below, and the one- / two-waves-per-transform kernels (ragged batches, coset)
            use_row2048 = true;
            test_ntt(log_n, 1, 0);
            test_ntt(log_n, 19, 0);
            test_coset(log_n, 3, 7);
            use_row2048 = false;
        }
        if (log_n >= 13 && log_n <= 15) {           // these sizes have two executors: first the two-pass plan ...
            use_lds = false;
            test_ntt(log_n, 3, 0);
            test_coset(log_n, 2, 7);
            use_lds = true;                         // ... then the single-sweep kernel in every workgroup shape, incl. ragged tiles
            for (lds_rows = 3; lds_rows <= 5; ++lds_rows) {
                test_ntt(log_n, ((1u << lds_rows) >> (log_n - 10)) + 1, 0);
                test_coset(log_n, 3, 7);
            }
            lds_rows = 5;
        }
        test_ntt(log_n, 1, 0);
        test_ntt(log_n, log_n <= 10 ? 70 : 3, 0);   // ragged row tiles for the single-pass kinds
        if (log_n == 8) test_ntt(log_n, 1, 1);      // src/ntt.rs:263-287 input
        test_coset(log_n, log_n <= 10 ? 5 : 2, 7);  // COSET_SHIFT = 7, src/fibonacci.rs:16
        if (log_n >= 11) {                          // every zero fraction the first pass of this plan supports
            NttPlan probe;
            build_plan(log_n, probe);
            for (int z = 1; z <= probe.pass[0].log_m; ++z) test_lde(log_n, z == 1 ? 3 : 1, z, z & 1 ? 7u : 1u);
        }
        if (log_n >= 1) {                           // Ext (AoS) transforms: one vector, a ragged few, plain and coset
            test_ext(log_n, 1, 1);
            test_ext(log_n, log_n <= 10 ? 11 : 2, 7);
            if (log_n >= 11) {
                NttPlan probe;
                build_plan(log_n, probe);
                for (int z = 1; z <= probe.pass[0].log_m; z += 2) test_ext(log_n, 1, 7, z);
            }
        }
        std::printf("log_n=%d failures=%d\n", log_n, failures);
        std::fflush(stdout);
    }
    for (int i = 2; i < argc; ++i) {                // extra sizes "LOG" or "LOGxBATCH" (2-pass 2^20, 3-pass 2^21.., wide tiles)
        const char* xb = std::strchr(argv[i], 'x');
        if (argv[i][0] == 'p') {                    // "pN": launches of <= 2^N 32-wide tiles take the three-step shapes from here on (-1: never)
            pass3_max_log_tiles32() = std::atoi(argv[i] + 1);
            continue;
        }
        if (argv[i][0] == 'Q') {
            latency_plan = argv[i][1] == '1';
            continue;
        }
        if (argv[i][0] == 'R') {                    // "R1" / "R0": n = 2^11 through the one-wave-per-transform kernel from here on
            use_row2048 = argv[i][1] == '1';
            continue;
        }
        if (argv[i][0] == 'b') {                    // "bN": vectors per "l" case from here on
            lde_batch = (uint64_t)std::atoll(argv[i] + 1);
            continue;
        }
        if (argv[i][0] == 'w') {                    // "wN": launches of >= 2^N 32-wide tiles take the 64-wide shapes of the 128/256-point passes
            wide_min_log_tiles32() = std::atoi(argv[i] + 1);
            continue;
        }
        if (argv[i][0] == 'l') {                    // "lLOGxZ": low-degree extension of 2^(LOG-Z) coefficients to 2^LOG points
            test_lde(std::atoi(argv[i] + 1), lde_batch, xb ? std::atoi(xb + 1) : 5, 7);
            std::printf("lde %s failures=%d\n", argv[i], failures);
            std::fflush(stdout);
            continue;
        }
        if (argv[i][0] == 'e') {                    // "eLOGxV": V Ext vectors of 2^LOG elements (AoS), plain + coset + LDE by 32 / 4
            const int lg = std::atoi(argv[i] + 1);
            const uint64_t vecs = xb ? (uint64_t)std::atoll(xb + 1) : 1;
            test_ext(lg, vecs, 1);
            test_ext(lg, vecs, 1234567891u);
            if (lg >= 11) { test_ext(lg, vecs, 7, 5); test_ext(lg, vecs, 7, 2); }
            std::printf("ext %s failures=%d\n", argv[i], failures);
            std::fflush(stdout);
            continue;
        }
        if (argv[i][0] == 's') {                    // "sLOGxG": one transform over G emulated ranks
            test_slab(std::atoi(argv[i] + 1), xb ? (uint64_t)std::atoll(xb + 1) : 2);
            std::printf("slab %s failures=%d\n", argv[i], failures);
            std::fflush(stdout);
            continue;
        }
        int log_n = std::atoi(argv[i]);
        test_ntt(log_n, xb ? (uint64_t)std::atoll(xb + 1) : 1, 0);
        if (!xb) test_coset(log_n, 1, 1234567891u);
        if (!xb) { test_lde(log_n, 1, 5, 7); test_lde(log_n, 1, 2, 7); }   // blowup 32 (COSET_SHIFT 7) and 4
        std::printf("log_n=%d failures=%d\n", log_n, failures);
        std::fflush(stdout);
    }
    for (int log_n : {13, 14, 16}) {
        if (log_n > max_log) continue;
        for (uint64_t G : {1, 2, 4}) {
            NttPlan probe;
            build_plan(log_n, probe);
            if (((1ull << log_n) >> probe.pass[0].log_m) / G < 32) continue;
            test_slab(log_n, G);
        }
        std::printf("slab log_n=%d failures=%d\n", log_n, failures);
    }
    for (int layer = 0; layer < 6; ++layer) test_fold(10, layer, 7);
    test_fold(13, 0, 7);
    test_fold(13, 12, 7);
    test_fold(1, 0, 7);
    test_fold(6, 2, 1);
    for (size_t n : {1, 2, 3, 4, 5, 8, 100, 1024}) { test_merkle(n, false); test_merkle(n, true); }
    test_merkle_coop();
    test_fold_xs_batch();
    test_prover_steps(6, 2, 7);
    test_prover_steps(9, 5, 7);
    test_prover_steps(8, 3, 1234567);
    for (size_t nc : {1, 15, 16, 17, 4095, 4096, 4097, 10000}) { test_poly_eval(nc, 987654321u); test_poly_eval(nc, 0u); test_poly_eval(nc, 1u); }
    for (size_t n : {1, 2, 3, 5, 8, 13, 100}) test_merkle_open(n);
    test_fold_ext(2);
    test_fold_ext(64);
    test_fold_ext(1024);
    std::printf("slab rows: fused %llu, two-step form %llu\n", slab_rows_fused, slab_rows_fallback);
    std::printf("tiles stepped: one-step %llu, two-step %llu, three-step %llu\n", tiles_by_steps[1], tiles_by_steps[2], tiles_by_steps[3]);
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ALL OK", failures);
    return failures ? 1 : 0;
}
