// The reference's three GPU tests (src/ntt.rs:253-311), restated in C++ over the host mirror
// toyni_amd/csrc/host/toyni_ntt.hpp; the CPU side (`cpu_ntt`) is the oracle.  Run by tests/test_gpu_cpp.py.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "../../toyni_amd/csrc/host/toyni_ntt.hpp"
#include "launch_dump.hpp"

extern "C" {
uint64_t orc_bb_new(uint64_t);
uint64_t orc_bb_root_of_unity(uint32_t);
int orc_ntt(uint64_t*, size_t, uint64_t);
int orc_fri_fold(uint64_t*, const uint64_t*, size_t, const uint64_t*, uint64_t);
int orc_domain_elements(uint64_t*, size_t, uint64_t);
uint64_t orc_bb_mul(uint64_t, uint64_t);
uint64_t orc_bb_add(uint64_t, uint64_t);
}

using toyni::BabyBear;
namespace ntt = toyni::ntt;

static int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { ++fails; std::printf("FAIL: " __VA_ARGS__); std::printf("\n"); } } while (0)

static void test_cuda_available() { std::printf("GPU available: %d\n", ntt::cuda_available()); }

static void test_cuda_ntt_vs_cpu() {
    if (!ntt::cuda_available()) { std::printf("GPU not available, skipping test\n"); return; }
    const size_t n = 256;
    std::vector<BabyBear> cpu_values(n);
    for (size_t i = 0; i < n; ++i) cpu_values[i] = {orc_bb_new(i * 7 + 3)};
    std::vector<BabyBear> gpu_values = cpu_values;
    const uint64_t omega = orc_bb_root_of_unity(8);
    orc_ntt(reinterpret_cast<uint64_t*>(cpu_values.data()), n, omega);
    ntt::ntt_cuda(gpu_values).unwrap();
    for (size_t i = 0; i < n; ++i)
        EXPECT(cpu_values[i].value == gpu_values[i].value, "Mismatch at index %zu: CPU=%llu, GPU=%llu", i,
               (unsigned long long)cpu_values[i].value, (unsigned long long)gpu_values[i].value);
}

static void test_cuda_intt_roundtrip() {
    if (!ntt::cuda_available()) { std::printf("GPU not available, skipping test\n"); return; }
    const size_t n = 256;
    std::vector<BabyBear> original(n);
    for (size_t i = 0; i < n; ++i) original[i] = {orc_bb_new(i * 7 + 3)};
    std::vector<BabyBear> values = original;
    ntt::ntt_cuda(values).unwrap();
    ntt::intt_cuda(values).unwrap();
    for (size_t i = 0; i < n; ++i) EXPECT(original[i].value == values[i].value, "Roundtrip failed at index %zu", i);
}

static void test_buffer_and_fold() {
    if (!ntt::cuda_available()) return;
    const size_t n = 64;
    std::vector<uint64_t> h(n), back(n);
    for (size_t i = 0; i < n; ++i) h[i] = i * 11 + 5;
    ntt::CudaBuffer buf(n);
    buf.copy_from_host(h).unwrap();
    buf.copy_to_host(back).unwrap();
    EXPECT(h == back, "CudaBuffer round trip");
    std::vector<BabyBear> evals(n), xs(n);
    std::vector<uint64_t> want(n / 2);
    for (size_t i = 0; i < n; ++i) evals[i] = {orc_bb_new(i * i * 977 + 13)};
    orc_domain_elements(reinterpret_cast<uint64_t*>(xs.data()), n, 7);
    orc_fri_fold(want.data(), reinterpret_cast<uint64_t*>(evals.data()), n, reinterpret_cast<uint64_t*>(xs.data()), 424242);
    auto got = toyni::fri_fold(evals, xs, {424242});
    for (size_t i = 0; i < n / 2; ++i) EXPECT(got[i].value == want[i], "fold mismatch at %zu", i);
    bool threw = false;
    try { std::vector<BabyBear> odd(3), x3(3); toyni::fri_fold(odd, x3, {1}); } catch (const std::logic_error&) { threw = true; }
    EXPECT(threw, "odd length must throw");
}

// src/math/domain.rs:220-242 (coset FFT == Horner at every coset point, shift 7, n = 8, 1 + 2x + 3x^2) and :193-218 (round trips)
static void test_domain_coset_fft_matches_horner() {
    if (!ntt::cuda_available()) return;
    toyni::BabyBearDomain domain = toyni::BabyBearDomain(8).with_gpu(true).get_coset({7});
    const std::vector<BabyBear> coeffs = {{1}, {2}, {3}};
    std::vector<BabyBear> evals = domain.fft(coeffs);
    std::vector<uint64_t> points(8);
    orc_domain_elements(points.data(), 8, 7);
    for (size_t i = 0; i < 8; ++i) {
        const uint64_t x = points[i];
        const uint64_t want = orc_bb_add(1, orc_bb_add(orc_bb_mul(2, x), orc_bb_mul(3, orc_bb_mul(x, x))));
        EXPECT(evals[i].value == want, "coset FFT mismatch at %zu", i);
    }
    std::vector<BabyBear> back = domain.ifft(evals);
    for (size_t i = 0; i < 8; ++i) EXPECT(back[i].value == (i < 3 ? coeffs[i].value : 0), "coset round trip at %zu", i);
    // a larger extension: 2^10 coefficients on a 2^15-point coset, back through ifft
    toyni::BabyBearDomain lde = toyni::BabyBearDomain(1 << 15).with_gpu(true).get_coset({7});
    std::vector<BabyBear> poly(1 << 10);
    for (size_t i = 0; i < poly.size(); ++i) poly[i] = {orc_bb_new(i * 2654435761ull + 17)};
    std::vector<BabyBear> rt = lde.ifft(lde.fft(poly));
    for (size_t i = 0; i < rt.size(); ++i) EXPECT(rt[i].value == (i < poly.size() ? poly[i].value : 0), "LDE round trip at %zu", i);
    bool threw = false;
    try { toyni::BabyBearDomain(8).fft(coeffs); } catch (const std::logic_error&) { threw = true; }
    EXPECT(threw, "the mirror has no CPU path");
}

// src/math/domain.rs:244-278: Ext FFT / IFFT round trip (n = 8, the reference's coefficients) and evaluation == Horner with
// coordinate-wise mul_base at every base domain point (3 coefficients, zero padding implied) -- plus a coset domain of 2^12 points
static void test_domain_ext_fft() {
    if (!ntt::cuda_available()) return;
    using toyni::Ext;
    toyni::BabyBearDomain domain = toyni::BabyBearDomain(8).with_gpu(true);
    std::vector<Ext> coeffs(8);
    for (uint64_t i = 0; i < 8; ++i) coeffs[i] = Ext{{{orc_bb_new(i * 3 + 1)}, {orc_bb_new(i + 2)}, {orc_bb_new(i * 7)}, {orc_bb_new(i + 5)}}};
    std::vector<Ext> evals = domain.fft_ext(coeffs);
    std::vector<Ext> rec = domain.ifft_ext(evals);
    for (size_t i = 0; i < 8; ++i)
        for (int k = 0; k < 4; ++k) EXPECT(rec[i].c[k].value == coeffs[i].c[k].value, "Ext FFT/IFFT roundtrip failed at %zu.%d", i, k);
    std::vector<Ext> three(3);
    for (uint64_t i = 0; i < 3; ++i) three[i] = Ext{{{orc_bb_new(i + 1)}, {orc_bb_new(i * 2)}, {orc_bb_new(i + 4)}, {orc_bb_new(7)}}};
    evals = domain.fft_ext(three);
    std::vector<uint64_t> xs(8);
    orc_domain_elements(xs.data(), 8, 1);
    for (size_t i = 0; i < 8; ++i)
        for (int k = 0; k < 4; ++k) {
            uint64_t acc = 0;
            for (int c = 2; c >= 0; --c) acc = orc_bb_add(orc_bb_mul(acc, xs[i]), three[c].c[k].value);   // acc.mul_base(x) + c, per coordinate
            EXPECT(evals[i].c[k].value == acc, "Ext FFT eval mismatch at %zu.%d", i, k);
        }
    toyni::BabyBearDomain lde = toyni::BabyBearDomain(1 << 12).with_gpu(true).get_coset({7});
    std::vector<Ext> poly(1 << 7);
    for (size_t i = 0; i < poly.size(); ++i) for (int k = 0; k < 4; ++k) poly[i].c[k] = {orc_bb_new(i * 40503ull + 17 * k + 1)};
    std::vector<Ext> back = lde.ifft_ext(lde.fft_ext(poly));
    bool ok = true;
    for (size_t i = 0; i < back.size(); ++i)
        for (int k = 0; k < 4; ++k) ok = ok && back[i].c[k].value == (i < poly.size() ? poly[i].c[k].value : 0ull);
    EXPECT(ok, "Ext coset LDE round trip");
}

// toyni_fri_commit_phase_device from compiled code: the fold loop of src/fibonacci.rs:222-245 with the transcript behind a C callback
// (what a Rust prover binds as an `extern "C" fn` trampoline over its FiatShamirTranscript).  Layers vs orc_fri_fold on the squared
// domain, trees vs orc_merkle_commit_values, and the callback must see exactly the committed roots, in order.
extern "C" {
size_t orc_merkle_total_digests(size_t);
void orc_merkle_commit_values(uint8_t*, const uint64_t*, const uint8_t*, size_t);
}
struct ToyTranscript {
    uint64_t state = 0x746f796e69ull;
    std::vector<std::vector<uint8_t>> roots;
    uint32_t squeeze() { state = state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)((state >> 24) % 2013265921ull); }
    void absorb(const uint8_t* r) { roots.emplace_back(r, r + 32); for (int i = 0; i < 32; ++i) state = (state ^ r[i]) * 1099511628211ull; }
};
static int toy_challenge(void* user, unsigned /*round*/, const uint8_t* prev_root, uint32_t* beta_out) {
    ToyTranscript* t = static_cast<ToyTranscript*>(user);
    if (prev_root) t->absorb(prev_root);
    if (beta_out) *beta_out = t->squeeze();
    return 0;
}
static void test_commit_phase_callback() {
    if (!ntt::cuda_available()) return;
    const size_t m0 = 256, final_size = 4;
    const uint32_t x0 = 7;
    std::vector<uint32_t> layer0(m0);
    for (size_t i = 0; i < m0; ++i) layer0[i] = (uint32_t)orc_bb_new(i * 2654435761ull + 99);
    size_t words = 0, digests = 0;
    for (size_t m = m0 / 2; m >= final_size; m /= 2) { words += m; digests += orc_merkle_total_digests(m); }
    uint64_t *d_in = nullptr, *d_layers = nullptr, *d_levels = nullptr;   // cuda_malloc counts u64 words
    EXPECT(cuda_malloc(&d_in, m0 / 2) == 0 && cuda_malloc(&d_layers, (words + 1) / 2) == 0 && cuda_malloc(&d_levels, digests * 4) == 0, "device allocation");
    EXPECT(cuda_copy_to_device(d_in, reinterpret_cast<const uint64_t*>(layer0.data()), m0 / 2) == 0, "upload");
    toyni_ntt_ctx* ctx = nullptr;
    EXPECT(toyni_ntt_ctx_create((uint32_t)m0, 0, &ctx) == 0, "context");
    ToyTranscript tr;
    unsigned rounds = 0;
    std::vector<uint8_t> roots(32 * 8);
    const int rc = toyni_fri_commit_phase_device(ctx, reinterpret_cast<const uint32_t*>(d_in), m0, x0, final_size, nullptr, toy_challenge, &tr,
                                                 reinterpret_cast<uint32_t*>(d_layers), reinterpret_cast<uint8_t*>(d_levels), roots.data(), &rounds, nullptr);
    EXPECT(rc == 0 && rounds == 6, "commit phase rc=%d rounds=%u", rc, rounds);
    std::vector<uint32_t> layers(words + 1);
    std::vector<uint8_t> levels(digests * 32);
    EXPECT(cuda_copy_from_device(reinterpret_cast<uint64_t*>(layers.data()), d_layers, (words + 1) / 2) == 0, "download layers");
    EXPECT(cuda_copy_from_device(reinterpret_cast<uint64_t*>(levels.data()), d_levels, digests * 4) == 0, "download trees");
    EXPECT(tr.roots.size() == rounds, "the callback absorbed %zu roots", tr.roots.size());
    // replay: same transcript, CPU fold, CPU trees
    ToyTranscript replay;
    std::vector<uint64_t> cur(layer0.begin(), layer0.end());
    uint64_t x = x0;
    size_t lo = 0, dlo = 0;
    for (unsigned k = 0; k < rounds; ++k) {
        if (k) replay.absorb(tr.roots[k - 1].data());
        const uint32_t beta = replay.squeeze();
        const size_t m = cur.size(), half = m / 2;
        std::vector<uint64_t> xs(m), want(half);
        orc_domain_elements(xs.data(), m, x);
        orc_fri_fold(want.data(), cur.data(), m, xs.data(), beta);
        for (size_t i = 0; i < half; ++i) EXPECT(layers[lo + i] == want[i], "round %u layer mismatch at %zu", k, i);
        const size_t nd = orc_merkle_total_digests(half);
        std::vector<uint8_t> tree(nd * 32);
        orc_merkle_commit_values(tree.data(), want.data(), nullptr, half);
        EXPECT(std::equal(tree.begin(), tree.end(), levels.begin() + dlo * 32), "round %u tree mismatch", k);
        EXPECT(std::equal(tr.roots[k].begin(), tr.roots[k].end(), tree.end() - 32) && std::equal(roots.begin() + 32 * k, roots.begin() + 32 * k + 32, tree.end() - 32),
               "round %u root", k);
        cur = want;
        x = orc_bb_mul(x, x);
        lo += half;
        dlo += nd;
    }
    toyni_ntt_ctx_destroy(ctx);
    cuda_free(d_in); cuda_free(d_layers); cuda_free(d_levels);
}

int main() {
    struct DumpAtReturn { ~DumpAtReturn() { toyni_test_dump_launched_kernels(); } } dump_at_return;   // kernel-coverage guard of the test session
    test_cuda_available();
    test_cuda_ntt_vs_cpu();
    test_cuda_intt_roundtrip();
    test_buffer_and_fold();
    test_domain_coset_fft_matches_horner();
    test_domain_ext_fft();
    test_commit_phase_callback();
    std::printf("%s\n", fails ? "CPP FAILED" : "CPP OK");
    return fails ? 1 : 0;
}
