// C++ host-side mirror of the reference's GPU wrapper module `src/ntt.rs::cuda` (src/ntt.rs:85-315),
// over the C ABI of include/toyni_hip.h.  The reference is compiled code (Rust) whose toolchain is
// absent here, so the host side above the ABI is written in C++ with the same names, argument
// meaning and error behaviour: Result<(), String> becomes toyni::Result (empty = Ok), assert! becomes
// an exception with the same message.
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "toyni_hip.h"  // include/toyni_hip.h (-I include)

namespace toyni {

// #[repr(C)] struct BabyBear { pub value: u64 }  (src/babybear.rs:10-14)
struct BabyBear {
    uint64_t value;
};
static_assert(sizeof(BabyBear) == sizeof(uint64_t) && alignof(BabyBear) == alignof(uint64_t), "layout (src/ntt.rs:115-116)");
// #[repr(C)] struct Ext { c: [BabyBear; 4] } = F_p[X]/(X^4 - 11)  (src/ext.rs): only the layout matters to the transforms
struct Ext {
    BabyBear c[4];
};
static_assert(sizeof(Ext) == 4 * sizeof(uint64_t), "Ext layout");

// Ok(()) / Err(String)
struct Result {
    std::string err;
    bool is_ok() const { return err.empty(); }
    void unwrap() const { if (!is_ok()) throw std::runtime_error(err); }
};

namespace ntt {

inline bool gpu_available() {  // src/ntt.rs:144-150
    int count = 0;
    return toyni_device_count(&count) == 0 && count > 0;
}

// src/ntt.rs:128-141: process-lifetime per-n cache
inline toyni_ntt_ctx* get_or_create_ctx(size_t n, std::string* err) {
    static std::mutex mu;
    static std::map<size_t, toyni_ntt_ctx*> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(n);
    if (it != cache.end()) return it->second;
    toyni_ntt_ctx* ctx = nullptr;
    int st = toyni_ntt_ctx_create((uint32_t)n, -1, &ctx);
    if (st != 0) { if (err) *err = std::string("GPU NTT context creation failed: ") + toyni_error_string(st); return nullptr; }
    cache[n] = ctx;
    return ctx;
}

inline Result transform(std::vector<BabyBear>& values, bool inverse) {
    if (!gpu_available()) return {"GPU not available"};                                     // src/ntt.rs:225-227
    const size_t n = values.size();
    if (n == 0 || (n & (n - 1))) throw std::logic_error("NTT size must be power of 2");     // src/ntt.rs:229
    if (n > (size_t(1) << 27)) throw std::logic_error("BabyBear only supports NTT up to 2^27");  // src/ntt.rs:230
    std::string err;
    toyni_ntt_ctx* ctx = get_or_create_ctx(n, &err);
    if (!ctx) return {err};
    int st = toyni_ntt_host(ctx, reinterpret_cast<uint64_t*>(values.data()), 1, inverse ? 1 : 0);
    if (st != 0) return {std::string("GPU NTT failed: ") + toyni_error_string(st)};
    return {};
}

inline Result ntt_gpu(std::vector<BabyBear>& values) { return transform(values, false); }   // src/ntt.rs:224-236
inline Result intt_gpu(std::vector<BabyBear>& values) { return transform(values, true); }   // src/ntt.rs:239-251

// src/ntt.rs:153-215; sizes in u64 elements
class GpuBuffer {
  public:
    explicit GpuBuffer(size_t size) : size_(size) {
        int st = cuda_malloc(&ptr_, size);
        if (st != 0) throw std::runtime_error(std::string("GPU malloc failed: ") + cuda_get_error_string(st));
    }
    GpuBuffer(const GpuBuffer&) = delete;
    GpuBuffer& operator=(const GpuBuffer&) = delete;
    ~GpuBuffer() { cuda_free(ptr_); }
    Result copy_from_host(const std::vector<uint64_t>& data) {
        if (data.size() != size_) throw std::logic_error("Size mismatch");
        int st = cuda_copy_to_device(ptr_, data.data(), size_);
        return st ? Result{std::string("GPU copy to device failed: ") + cuda_get_error_string(st)} : Result{};
    }
    Result copy_to_host(std::vector<uint64_t>& data) const {
        if (data.size() != size_) throw std::logic_error("Size mismatch");
        int st = cuda_copy_from_device(data.data(), ptr_, size_);
        return st ? Result{std::string("GPU copy from device failed: ") + cuda_get_error_string(st)} : Result{};
    }
    uint64_t* as_ptr() const { return ptr_; }

  private:
    uint64_t* ptr_ = nullptr;
    size_t size_;
};

// the reference's names (feature `cuda`, src/ntt.rs:314-315)
inline bool cuda_available() { return gpu_available(); }
inline Result ntt_cuda(std::vector<BabyBear>& v) { return ntt_gpu(v); }
inline Result intt_cuda(std::vector<BabyBear>& v) { return intt_gpu(v); }
using CudaBuffer = GpuBuffer;

}  // namespace ntt

// The caller of the hot path, BabyBearDomain (src/math/domain.rs:10-175), restricted to what reaches the GPU: fft / ifft with
// `use_gpu` set.  Zero padding, coset scaling and the transform are one ABI call each (toyni_lde_host / toyni_coset_ntt_host);
// the reference pads and scales in serial host loops first (:108-111, :154-174).  Without `with_gpu(true)` this mirror throws:
// the CPU transform is the reference's own src/ntt.rs.
class BabyBearDomain {
  public:
    explicit BabyBearDomain(size_t size) : size_(size) {
        if (size == 0 || (size & (size - 1))) throw std::logic_error("Domain size must be power of 2");   // src/math/domain.rs:21
    }
    BabyBearDomain get_coset(BabyBear shift) const { BabyBearDomain d(size_); d.shift_ = shift.value % 2013265921ull; d.use_gpu_ = use_gpu_; return d; }  // :34-42
    BabyBearDomain& with_gpu(bool use_gpu) { use_gpu_ = use_gpu; return *this; }                         // :45-48
    size_t size() const { return size_; }

    std::vector<BabyBear> fft(const std::vector<BabyBear>& coeffs) const {                               // :107-123
        toyni_ntt_ctx* ctx = context();
        std::vector<BabyBear> out(size_);
        const size_t take = coeffs.size() < size_ ? coeffs.size() : size_;                               // `resize` truncates, :109
        int st = toyni_lde_host(ctx, reinterpret_cast<const uint64_t*>(coeffs.data()), take, reinterpret_cast<uint64_t*>(out.data()), shift_);
        if (st != 0) throw std::runtime_error(std::string("GPU NTT failed: ") + toyni_error_string(st));  // `expect`, :116
        return out;
    }
    std::vector<BabyBear> ifft(const std::vector<BabyBear>& evals) const {                               // :85-102
        if (evals.size() != size_) throw std::logic_error("Evaluation count must match domain size");    // :86
        toyni_ntt_ctx* ctx = context();
        std::vector<BabyBear> values = evals;
        int st = toyni_coset_ntt_host(ctx, reinterpret_cast<uint64_t*>(values.data()), 1, shift_, 1);
        if (st != 0) throw std::runtime_error(std::string("GPU INTT failed: ") + toyni_error_string(st));
        return values;
    }

    // fft_ext / ifft_ext (:129-151): the reference de-interleaves the four coordinates, transforms each and re-interleaves; here the
    // AoS vector (Ext = #[repr(C)] [BabyBear; 4], src/ext.rs) goes through the interleaved passes as it is -- one ABI call each.
    std::vector<Ext> fft_ext(const std::vector<Ext>& coeffs) const {
        toyni_ntt_ctx* ctx = context();
        std::vector<Ext> out(size_);
        const size_t take = coeffs.size() < size_ ? coeffs.size() : size_;                               // `resize` truncates, :137
        int st = toyni_lde_ext_host(ctx, reinterpret_cast<const uint64_t*>(coeffs.data()), take, reinterpret_cast<uint64_t*>(out.data()), shift_);
        if (st != 0) throw std::runtime_error(std::string("GPU Ext NTT failed: ") + toyni_error_string(st));
        return out;
    }
    std::vector<Ext> ifft_ext(const std::vector<Ext>& evals) const {
        if (evals.size() != size_) throw std::logic_error("Evaluation count must match domain size");    // :130
        toyni_ntt_ctx* ctx = context();
        std::vector<Ext> values = evals;
        int st = toyni_ntt_ext_host(ctx, reinterpret_cast<uint64_t*>(values.data()), shift_, 1);
        if (st != 0) throw std::runtime_error(std::string("GPU Ext INTT failed: ") + toyni_error_string(st));
        return values;
    }

  private:
    toyni_ntt_ctx* context() const {
        if (!use_gpu_) throw std::logic_error("this mirror ships the GPU path only (with_gpu(true)); the CPU transform is src/ntt.rs");
        if (!ntt::gpu_available()) throw std::runtime_error("GPU not available");
        std::string err;
        toyni_ntt_ctx* ctx = ntt::get_or_create_ctx(size_, &err);
        if (!ctx) throw std::runtime_error(err);
        return ctx;
    }
    size_t size_;
    uint64_t shift_ = 1;
    bool use_gpu_ = false;
};

// src/math/fri.rs:27-48
inline std::vector<BabyBear> fri_fold(const std::vector<BabyBear>& evals, const std::vector<BabyBear>& xs, BabyBear beta) {
    if (evals.size() % 2) throw std::logic_error("Evaluations length must be even");
    std::vector<BabyBear> out(evals.size() / 2);
    int st = toyni_fri_fold_host(reinterpret_cast<uint64_t*>(out.data()), reinterpret_cast<const uint64_t*>(evals.data()), evals.size(),
                                 reinterpret_cast<const uint64_t*>(xs.data()), beta.value);
    if (st == TOYNI_E_ZERO_INVERSE) throw std::logic_error("Cannot invert zero");
    if (st != 0) throw std::runtime_error(std::string("GPU FRI fold failed: ") + toyni_error_string(st));
    return out;
}

}  // namespace toyni
