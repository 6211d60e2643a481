// The reference's three GPU tests (src/ntt.rs:253-311), restated in C++ over the host mirror
// toyni_amd/csrc/host/toyni_ntt.hpp; the CPU side (`cpu_ntt`) is the oracle.  Run by tests/test_gpu_cpp.py.
#include <cstdio>
#include <vector>

#include "../../toyni_amd/csrc/host/toyni_ntt.hpp"

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

int main() {
    test_cuda_available();
    test_cuda_ntt_vs_cpu();
    test_cuda_intt_roundtrip();
    test_buffer_and_fold();
    test_domain_coset_fft_matches_horner();
    std::printf("%s\n", fails ? "CPP FAILED" : "CPP OK");
    return fails ? 1 : 0;
}
