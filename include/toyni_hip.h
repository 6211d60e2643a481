/*
 * toyni_hip.h -- C ABI of libtoyni_hip.so: the MI355X (gfx950) backend for Toyni's BabyBear NTT and
 * FRI fold.  This is the drop-in boundary: it is what `src/ntt.rs::cuda` (reference
 * /root/reference/src/ntt.rs:95-110) binds, re-implemented from scratch in HIP.  Plain pointers and
 * sizes only.  Every new entry point returns an int status (0 = success, otherwise a hipError_t value
 * or one of the TOYNI_E_* codes below; toyni_error_string() names both).
 *
 * Element formats
 *   host / "u64" : one uint64_t per element, canonical residue mod p = 2013265921
 *                  (= #[repr(C)] struct BabyBear { value: u64 }, src/babybear.rs:10-14)
 *   device "u32" : packed uint32_t canonical residues (the native device format; 8 B/element of HBM
 *                  traffic per pass instead of 16).
 *
 * Threading and streams: a context serialises the ENQUEUEING of its calls with an internal mutex, and keeps its
 * intermediate buffers per stream: calls enqueued on one stream are ordered by that stream, calls enqueued on different
 * streams use different buffers.  So one context may be driven from several host threads and several streams at once
 * (the reference's single shared d_data -- SURVEY.md F8, cuda/ntt_kernel.cu:205 -- has no counterpart).  At most 8 streams'
 * buffer sets are kept per context (least recently used first out).  Memory stays bounded without any call from the user
 * (round 3): a buffer that was outgrown, or that belonged to an evicted set, carries a HIP event recorded on its stream after its
 * last use, and the next call on the context that finds the event complete frees it (hipFree synchronises the device, so this
 * never happens on the enqueue path of a steady-state caller -- there is nothing to free there).  A context that serves ONE stream
 * with small calls (intermediates under 64 MiB: the latency path) records no events at all; with several streams, or large
 * intermediates, every call ends with one hipEventRecord (TOYNI_FENCE=always|never overrides).
 * Synchronising a stream through this API (blocking entry points, toyni_stream_synchronize), toyni_ntt_ctx_trim and
 * toyni_ntt_ctx_destroy release retired buffers as before.  Calls captured into a HIP graph leave no fence.
 */
#ifndef TOYNI_HIP_H
#define TOYNI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOYNI_OK 0
#define TOYNI_E_INVALID_SIZE 10001   /* n not a power of two, or > 2^27 (src/ntt.rs:229-230) */
#define TOYNI_E_NULL 10002           /* null context / pointer */
#define TOYNI_E_ODD_LENGTH 10003     /* fri_fold: "Evaluations length must be even" (src/math/fri.rs:28) */
#define TOYNI_E_NO_DEVICE 10004      /* no usable GPU ("CUDA not available", src/ntt.rs:225-227) */
#define TOYNI_E_ZERO_INVERSE 10005   /* fri_fold: a point x_i = 0 ("Cannot invert zero", src/babybear.rs:112) */
#define TOYNI_E_RANGE 10006          /* argument out of range (layer larger than the context's domain, ...) */
#define TOYNI_E_NO_RCCL 10007        /* the RCCL exchange was requested and librccl could not be loaded */
#define TOYNI_E_RCCL 10008           /* an RCCL call failed */
#define TOYNI_E_NO_PEER_ACCESS 10009 /* two devices of a multi-device group cannot access each other directly (hipDeviceCanAccessPeer);
                                      * TOYNI_ALLOW_STAGED_PEER=1 accepts host-staged copies instead */
#define TOYNI_E_REENTRANT 10011      /* an entry point was called on a context from inside that context's transcript callback */
#define TOYNI_E_SELF_CHECK 10010     /* the first-use check of a multi-device group disagreed with the single-device transform */

typedef struct toyni_ntt_ctx toyni_ntt_ctx;

/* ------------------------------------------------------------------------------------------------
 * 1. The reference's ABI, symbol for symbol (cuda/ntt_kernel.cu:211-318; extern block src/ntt.rs:95-110).
 *    The reference's own extern block binds all ten of its symbols with ONE edit: the link name
 *    (`ntt_cuda` -> `toyni_hip`).
 *    `count` is in u64 ELEMENTS.
 * ---------------------------------------------------------------------------------------------- */
void* ntt_ctx_create(uint32_t n);                       /* cuda/ntt_kernel.cu:213-234; NULL on error */
void ntt_ctx_destroy(void* ctx);                        /* :236-242; null-safe */
void ntt_run_inplace(void* ctx, uint64_t* h_data);      /* :249-268; host pointer, ctx->n elements, blocking */
void intt_run_inplace(void* ctx, uint64_t* h_data);     /* :272-292 */
int cuda_malloc(uint64_t** d_ptr, size_t count);        /* :298-300 */
int cuda_free(uint64_t* d_ptr);                         /* :302-304 */
int cuda_copy_to_device(uint64_t* d_dest, const uint64_t* h_src, size_t count);    /* :306-308 */
int cuda_copy_from_device(uint64_t* h_dest, const uint64_t* d_src, size_t count);  /* :310-312 */
const char* cuda_get_error_string(int error);           /* :314-316 */

/* The tenth symbol of the reference's extern block, `cudaGetDeviceCount` (src/ntt.rs:102,147), comes from libcudart there.  It is
 * exported here under the same name and signature (count of HIP devices, 0 = success), so the reference's extern block links against
 * this library with ONE edit, the link name (`ntt_cuda` -> `toyni_hip`); toyni_device_count is the same call under a neutral name,
 * which rust/src/ntt_gpu.rs binds. */
int cudaGetDeviceCount(int* count);
int toyni_device_count(int* count);
const char* toyni_error_string(int status);

/* ------------------------------------------------------------------------------------------------
 * 2. Status-returning context API (superset of section 1)
 * ---------------------------------------------------------------------------------------------- */
/* n: power of two, 1 <= n <= 2^27.  device: HIP device ordinal, or -1 for the current device.
 * Builds both directions' twiddle tables (cuda/ntt_kernel.cu:160-185,213-234) and owns a stream. */
int toyni_ntt_ctx_create(uint32_t n, int device, toyni_ntt_ctx** out);
int toyni_ntt_ctx_destroy(toyni_ntt_ctx* ctx);
uint32_t toyni_ntt_ctx_n(const toyni_ntt_ctx* ctx);
int toyni_ntt_ctx_device(const toyni_ntt_ctx* ctx);
int toyni_ntt_ctx_passes(const toyni_ntt_ctx* ctx);     /* HBM sweeps per transform of the context's plan (1..3); large batches
                                                          * of n = 2^11..2^13 run a single-sweep kernel instead of their 2 passes */
/* HBM sweeps (= kernel launches) one toyni_ntt_device call on `batch` base-field transforms makes: the single-sweep kernel for large
 * batches of n = 2^11..2^13, the two-pass plan of n = 2^21 (every batch) and of a lone n = 2^22 transform, else as above. */
int toyni_ntt_ctx_passes_for(const toyni_ntt_ctx* ctx, size_t batch);
/* Multi-pass transforms of a large batch are issued in chunks of about chunk_elems elements so that the
 * intermediate buffer stays cache-resident; 0 = whole batch at once (also env TOYNI_CHUNK_ELEMS). */
int toyni_ntt_ctx_set_chunk(toyni_ntt_ctx* ctx, size_t chunk_elems);

/* Host-slice entry points -- what ntt_cuda / intt_cuda (src/ntt.rs:224-251) call.  h_data: batch * n
 * u64 elements, overwritten in place, natural order in and out, canonical root w_n.  Blocking.
 * H2D -> narrow -> fused passes -> widen -> D2H. */
int toyni_ntt_host(toyni_ntt_ctx* ctx, uint64_t* h_data, size_t batch, int inverse);

/* The same over several GPUs from ONE host process: the batch is sharded contiguously over `devices` (ordinals; a
 * device may be listed more than once), one host thread per entry, no collective.  Contexts are cached per
 * (device, lane, n) for the life of the process, like get_or_create_ctx (src/ntt.rs:128-141).  Blocking. */
int toyni_ntt_host_multi_gpu(const int* devices, int ndev, uint32_t n, uint64_t* h_data, size_t batch, int inverse);

/* Device-resident, packed u32, in place (d_in == d_out) or out of place.  Enqueued on `stream`
 * (a hipStream_t; NULL = HIP's default stream, as everywhere in HIP) and NOT synchronised: this is the
 * entry point the roofline numbers are measured on.  batch transforms are contiguous (stride n); the pointers need
 * 4-byte alignment only; elements are canonical residues (< p), outputs are canonical.
 * Any number of streams per context (intermediates are per stream, see "Threading and streams" above). */
int toyni_ntt_device(toyni_ntt_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t batch, int inverse, void* stream);

/* Device-resident on the reference's u64 element layout (what a CudaBuffer holds, src/ntt.rs:153-215). */
int toyni_ntt_device_u64(toyni_ntt_ctx* ctx, uint64_t* d_data, size_t batch, int inverse, void* stream);

/* Coset transforms of BabyBearDomain (src/math/domain.rs:85-123,154-174) with the host's serial
 * shift^i loop fused on the device: forward = scale by shift^i then NTT; inverse = INTT then scale by
 * shift^-i.  shift is a canonical nonzero residue; shift == 1 is the plain transform. */
int toyni_coset_ntt_device(toyni_ntt_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t batch, uint32_t shift, int inverse, void* stream);
int toyni_coset_ntt_host(toyni_ntt_ctx* ctx, uint64_t* h_data, size_t batch, uint64_t shift, int inverse);

/* Low-degree extension: the forward coset transform of `batch` coefficient vectors of n >> log_blowup words each
 * (contiguous, stride n >> log_blowup), zero-padded to n -- what the prover does with every column (src/fibonacci.rs:101-103;
 * BabyBearDomain::fft pads, src/math/domain.rs:107-123).  The padding is implied, not stored: the first pass reads only the
 * words that exist and skips the butterflies whose partner is a padding zero.  Out of place; d_out holds batch * n words.
 * Identical results to zero-padding by hand and calling toyni_coset_ntt_device. */
int toyni_lde_device(toyni_ntt_ctx* ctx, const uint32_t* d_coeffs, uint32_t* d_out, size_t batch, unsigned log_blowup, uint32_t shift, void* stream);
/* Host-slice form = BabyBearDomain::fft(coeffs) in one call (src/math/domain.rs:107-123): ncoeffs <= n coefficients in
 * (any count; u64 elements, reduced mod p like BabyBear::new), n evaluations on shift * <w_n> out.  Only the
 * coefficients are uploaded (the reference pads on the host and uploads n elements).  Blocking. */
int toyni_lde_host(toyni_ntt_ctx* ctx, const uint64_t* h_coeffs, size_t ncoeffs, uint64_t* h_out, uint64_t shift);

/* Extension-field transforms, fft_ext / ifft_ext (src/math/domain.rs:129-151): n Ext elements = 4 words each (AoS,
 * #[repr(C)] Ext { c: [BabyBear; 4] }).  The transform is base-linear, i.e. the base transform of every coordinate (the reference
 * de-interleaves into four vectors, transforms each and re-interleaves, :140-151).  Here the AoS vector is transformed AS IT IS: the
 * pass kernels have an interleaved form in which the four coordinates of an element are four neighbouring columns (first / middle
 * passes) or four interleaved rows (last pass), so an Ext transform costs exactly the passes of the plan -- no de-interleave sweep
 * on either side -- and `batch` vectors go through one launch sequence.  shift = coset shift (1 = standard domain).
 * _batch_device: d_in == d_out allowed; batch * n * 4 words each. */
int toyni_ntt_ext_host(toyni_ntt_ctx* ctx, uint64_t* h_data, uint64_t shift, int inverse);
int toyni_ntt_ext_device(toyni_ntt_ctx* ctx, uint32_t* d_data, uint32_t shift, int inverse, void* stream);
int toyni_ntt_ext_batch_device(toyni_ntt_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t batch, uint32_t shift, int inverse, void* stream);

/* fft_ext of a coefficient vector shorter than the domain (src/math/domain.rs:134-151 pads every coordinate column): the interleaved
 * low-degree extension, padding implied (see toyni_lde_device).  Ext elements are 4 words each (AoS) on both sides.
 * host: ncoeffs <= n Ext coefficients (4 * ncoeffs u64) in, n Ext evaluations (4 * n u64) out, only the coefficients uploaded;
 * device: `batch` vectors of (n >> log_blowup) Ext coefficients (packed u32 AoS) in, n Ext evaluations each out, out of place. */
int toyni_lde_ext_host(toyni_ntt_ctx* ctx, const uint64_t* h_coeffs, size_t ncoeffs, uint64_t* h_out, uint64_t shift);
int toyni_lde_ext_device(toyni_ntt_ctx* ctx, const uint32_t* d_coeffs, uint32_t* d_out, unsigned log_blowup, uint32_t shift, void* stream);
int toyni_lde_ext_batch_device(toyni_ntt_ctx* ctx, const uint32_t* d_coeffs, uint32_t* d_out, size_t batch, unsigned log_blowup, uint32_t shift, void* stream);

/* Multi-GPU 4-step transform of one size-n vector (n = n1 * n2 over G ranks, one all-to-all): the twiddle between
 * the two local stages, d_data[r][k] *= w_n^(+-(row0 + r) * k) for r < rows, k < row_len (ctx of size n;
 * (row0 + rows) * row_len <= n).  The local stages are toyni_ntt_device batches; the exchange is the caller's
 * RCCL all-to-all (toyni_amd/dist.py). */
int toyni_fourstep_twiddle_device(toyni_ntt_ctx* ctx, uint32_t* d_data, size_t rows, size_t row_len, size_t row0, int inverse, void* stream);

/* 2b. One size-n transform over G devices WITHOUT local transposes (one process per GPU; the exchange is the caller's
 * all-to-all).  View x as [M1][S1], M1 = toyni_ntt_ctx_first_pass_points(ctx) (0 when n <= 1024: nothing to split),
 * S1 = n / M1.  The transform's first pass couples only elements of one column, so a rank that owns a block of
 * columns runs it alone (the same strided pass kernel the single-device transform uses); what remains for every k1
 * is an ordinary size-S1 transform (toyni_ntt_device on a size-S1 context):
 *   forward: slab pass (twiddle fused) -> all-to-all of contiguous row blocks -> relayout(0) -> size-S1 row transforms
 *   inverse: size-S1 inverse row transforms -> relayout(1) (twiddle fused) -> all-to-all -> slab pass(inverse)
 * Four HBM sweeps per direction at n = 2^27 instead of the eight of the transpose-based 4-step above -- three with
 * toyni_ntt_slab_rows_device, which folds the relayout into the row transforms' addressing.
 * Layouts: forward input / inverse output = slab [M1][cols_local], element (j1, c) = x[j1 S1 + col_base + c];
 *          forward output / inverse input = rows [rows_local][S1], element (r, k') = X[(row0 + r) + M1 k'].
 * cols_local and rows_local are powers of two, cols_local >= 32.  (src/ntt.rs:11-81 has no multi-device form: values
 * are pinned by the single-device transform and the oracle on the gathered result.) */
size_t toyni_ntt_ctx_first_pass_points(const toyni_ntt_ctx* ctx);
size_t toyni_first_pass_points(uint32_t n);   /* the same from n alone: no context, no device (0: n <= 1024 or not a valid size) */
/* in place on the slab.  inverse = 0: M1-point column transforms times w_n^((col_base + c) k1);
 * inverse = 1: the closing inverse column transforms, scaled by 1/M1 (col_base is not used: the twiddle was applied
 * by the relayout on the other side of the exchange) */
int toyni_ntt_slab_pass_device(toyni_ntt_ctx* ctx, uint32_t* d_slab, size_t cols_local, size_t col_base, int inverse, void* stream);
/* W = S1 / parts.  inverse = 0 (after the exchange):  in [parts][rows_local][W] -> out [rows_local][parts][W] = [rows_local][S1];
 * inverse = 1 (before the exchange): in [rows_local][S1] -> out [parts][rows_local][W], times w_n^-((row0 + r) j').
 * Out of place (d_in != d_out). */
int toyni_ntt_slab_relayout_device(toyni_ntt_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t rows_local, size_t row0,
                                   size_t parts, int inverse, void* stream);
/* The relayout and the size-S1 row transforms as ONE step either side of the exchange (round 5), `row_ctx` = a context of size S1 on
 * the same device:
 *   inverse = 0: d_in = the received pieces [parts][rows_local][W]  ->  d_out = rows [rows_local][S1], transformed
 *   inverse = 1: d_in = rows [rows_local][S1]  ->  d_out = pieces, inverse-transformed (1/S1) and times w_n^-((row0 + r) j')
 * Where the pass shapes the launch takes can address the pieces layout, no relayout sweep exists: the first pass of the forward row
 * transforms reads the pieces, the last pass of the inverse ones writes them -- three HBM sweeps per direction around the exchange
 * instead of four; otherwise (a few rows, rows of at most 1024 points, a chunked context) the two steps above run one after the
 * other.  Same results either way; *fused (may be NULL) reports which ran.  Out of place; the inverse form may overwrite d_in. */
int toyni_ntt_slab_rows_device(toyni_ntt_ctx* ctx, toyni_ntt_ctx* row_ctx, uint32_t* d_in, uint32_t* d_out, size_t rows_local, size_t row0,
                               size_t parts, int inverse, int* fused, void* stream);

/* 2c. The same transform driven from ONE host process over G = ndev devices (G a power of two; Toyni is a single process,
 * src/ntt.rs:128-141): slab pass -> ONE exchange -> relayout -> row transforms on every device, contexts, streams and
 * landing buffers cached per (device list, n).  `exchange` selects how the one exchange step moves its G x G blocks:
 *   TOYNI_EXCHANGE_PEER_COPY  every destination pulls its blocks with hipMemcpyPeerAsync (xGMI), one copy stream per
 *                             source so that all incoming links are busy at once; the only form that accepts a device
 *                             listed more than once (lanes on one device: the copy is then device-local).  The environment
 *                             variable TOYNI_SLAB_PIECES = K (power of two, default 1) issues the exchange as K pieces of every
 *                             row block, so that the work on piece q overlaps the transfer of the pieces after it
 *   TOYNI_EXCHANGE_RCCL       one ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd over ncclCommInitAll
 *                             communicators; librccl is loaded on first use (dlopen), TOYNI_E_NO_RCCL if it is absent
 * Both entry points block until the transform is complete on every device -- on error paths too: no stream of the group still
 * touches the caller's buffers when they return.  Peer access between the listed devices is checked (TOYNI_E_NO_PEER_ACCESS), and the
 * first call on a device list that spans more than one device runs one n = 2^18 transform through the same exchange and compares
 * it with the single-device result (TOYNI_E_SELF_CHECK on a mismatch; a few ms, once per device list and exchange kind).
 * TOYNI_VERBOSE=1 prints the lane -> device / PCI bus table and the self-check's verdict to stderr.
 * _device: d_slabs[g] = device g's [M1][S1/G] column slab, d_rows[g] = its [M1/G][S1] row block (layouts of 2b, packed u32,
 *          resident on devices[g]).  forward: slabs in (OVERWRITTEN), rows out; inverse: rows in (OVERWRITTEN), slabs out.
 * _host:   n u64 elements in natural order, in place (x -> X or X -> x); every lane uploads / downloads its own strided share. */
#define TOYNI_EXCHANGE_PEER_COPY 0
#define TOYNI_EXCHANGE_RCCL 1
int toyni_ntt_slab_multi_gpu_device(const int* devices, int ndev, uint32_t n, uint32_t* const* d_slabs, uint32_t* const* d_rows,
                                    int inverse, int exchange);
int toyni_ntt_slab_multi_gpu_host(const int* devices, int ndev, uint32_t n, uint64_t* h_data, int inverse, int exchange);

/* Domain points on the device: d_out[i] = shift * w_m^i, i < m (roots_of_unity_domain, src/ntt.rs:69-81, with shift = 1;
 * BabyBearDomain::elements, src/math/domain.rs:61-69) -- the xs that fri_fold and the prover's pointwise steps consume.
 * m: power of two <= the context's n.  The reference's serial multiply chain becomes independent table lookups. */
int toyni_domain_elements_device(toyni_ntt_ctx* ctx, uint32_t* d_out, size_t m, uint32_t shift, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3. FRI pairwise fold (net-new on the device; oracle src/math/fri.rs:27-48)
 *    out[i] = (a + b)/2 + (a - b)/2 * beta / x_i,  a = evals[i], b = evals[i + m/2],  i < m/2
 * ---------------------------------------------------------------------------------------------- */
/* Structured points x_i = x0 * w_m^i (the prover's layers: x0 = shift^(2^k), src/fibonacci.rs:214,228-231).
 * ctx: any context whose n >= m (its inverse-root table is used).  m even power of two >= 2.  Async on stream. */
int toyni_fri_fold_device(toyni_ntt_ctx* ctx, const uint32_t* d_evals, uint32_t* d_out, size_t m, uint32_t beta, uint32_t x0, void* stream);

/* The prover's whole fold loop (src/fibonacci.rs:222-245) without the Merkle commits: nfolds layers of
 * a size-n codeword on the coset shift * <w_n>; layer k (size n >> (k+1)) is written at
 * d_layers + (n - (n >> k)) ... i.e. back to back: n/2, n/4, ...  betas: host array of nfolds challenges. */
int toyni_fri_fold_layers_device(toyni_ntt_ctx* ctx, const uint32_t* d_evals, uint32_t* d_layers, const uint32_t* betas, unsigned nfolds, uint32_t shift, void* stream);

/* Explicit points, the reference's signature fri_fold(evals, xs, beta): only xs[0 .. m/2) is read. */
int toyni_fri_fold_xs_device(const uint32_t* d_evals, const uint32_t* d_xs, uint32_t* d_out, size_t m, uint32_t beta, void* stream);
/* Host-slice form of the same (u64 elements, blocking): out has len/2 elements. */
int toyni_fri_fold_host(uint64_t* h_out, const uint64_t* h_evals, size_t len, const uint64_t* h_xs, uint64_t beta);

/* Extension-field codewords, fri_fold_ext (src/math/fri.rs:7-25): values and beta in Ext = F_p[X]/(X^4 - 11)
 * (src/ext.rs), points in the base field.  Elements are AoS: 4 consecutive words c0..c3 (u32 on the device, u64 on
 * the host = #[repr(C)] Ext { c: [BabyBear; 4] }).  m / len count ELEMENTS.  beta = 4 canonical coordinates. */
int toyni_fri_fold_ext_device(toyni_ntt_ctx* ctx, const uint32_t* d_evals, uint32_t* d_out, size_t m, const uint32_t beta[4], uint32_t x0, void* stream);
int toyni_fri_fold_ext_xs_device(const uint32_t* d_evals, const uint32_t* d_xs, uint32_t* d_out, size_t m, const uint32_t beta[4], void* stream);
int toyni_fri_fold_ext_host(uint64_t* h_out, const uint64_t* h_evals, size_t len, const uint64_t* h_xs, const uint64_t beta[4]);

/* ------------------------------------------------------------------------------------------------
 * 3b. Merkle commitment of a layer (oracle: src/merkle.rs:25-48,105-123 and the leaf format of
 *     build_merkle_tree / build_unsalted_tree, src/fibonacci.rs:340-361).  leaf = SHA256(0x00 || salt[16] || value as 8 LE
 *     bytes) -- or 0x00 || value bytes when salts == NULL; node = SHA256(0x01 || left || right); odd levels duplicate
 *     their last node.  Output: ALL levels back to back as 32-byte digests (leaf hashes first, root last) =
 *     MerkleTree::levels; toyni_merkle_total_digests(n) digests.  d_levels / d_salts 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
size_t toyni_merkle_total_digests(size_t n);
int toyni_merkle_commit_device(const uint32_t* d_values, const uint8_t* d_salts, size_t n, uint8_t* d_levels, void* stream);
int toyni_merkle_commit_host(const uint64_t* h_values, const uint8_t* h_salts, size_t n, uint8_t* h_levels);

/* ------------------------------------------------------------------------------------------------
 * 3c. One FRI round, and the pointwise steps of the Fibonacci prover on the LDE coset (oracle:
 *     src/fibonacci.rs:133-150,186-198,222-245, src/math/polynomial.rs:134-144, src/merkle.rs:50-80).  All asynchronous on
 *     `stream`; packed u32 device data; ctx = a context of size N = the LDE size (its domain table supplies x_i = shift w_N^i).
 * ---------------------------------------------------------------------------------------------- */
/* One round of the fold loop (src/fibonacci.rs:222-245): d_out = fri_fold(d_evals) on the points x0 w_m^i with challenge
 * beta, AND the Merkle tree of the folded layer in d_levels (layout of toyni_merkle_commit_device over m/2 leaves; the leaf
 * hashes are computed in the fold's own sweep; d_salts = (m/2) x 16 bytes, or NULL for the unsalted final layer).  The next
 * beta depends on this tree's root (the last 32 bytes of d_levels), so a round is as far as the protocol lets the fusion go. */
int toyni_fri_fold_commit_device(toyni_ntt_ctx* ctx, const uint32_t* d_evals, uint32_t* d_out, size_t m, uint32_t beta, uint32_t x0,
                                 const uint8_t* d_salts, uint8_t* d_levels, void* stream);
/* The WHOLE fold loop of the commit phase (src/fibonacci.rs:222-245) in one blocking call: rounds of toyni_fri_fold_commit_device
 * while the layer is longer than final_size (a power of two, >= 1).  The protocol's dependency is kept, not broken: before every
 * fold the library calls `challenge(user, round, root, &beta)` with the root committed by the previous round (root = NULL in
 * round 0) -- the caller's Fiat-Shamir transcript absorbs it (absorb_commitment, src/transcript.rs:29-32) and squeezes beta
 * (squeeze_challenge, :34-40); after the last round it is called once more with beta_out = NULL to absorb the last root.  A
 * non-zero return of the callback aborts with that code (after the stream has drained).  The callback runs on the calling thread
 * with the context locked: entry points on the SAME context return TOYNI_E_REENTRANT from inside it (other contexts are fine).
 * Per round only the 32-byte root crosses PCIe.
 *   d_layer0 : m0 words on the points x0 w_m0^i (the DEEP layer, :205-214); round k folds on x0^(2^k) (the squared domain, :228-231)
 *   d_salts  : 16 bytes per leaf for every SALTED layer back to back (m0/2 + m0/4 + ... leaves, the final layer excluded:
 *              build_unsalted_tree, :236-240), or NULL for unsalted trees throughout
 *   d_layers : out, the folded layers back to back (m0/2 + m0/4 + ... + final_size words)
 *   d_levels : out, their trees back to back (toyni_merkle_total_digests(m_k) x 32 bytes each, 16-byte aligned)
 *   h_roots  : out, rounds x 32 bytes (may be NULL); *rounds_out = number of rounds = log2(m0 / final_size)
 * `stream` carries the kernels; the call returns after the last root has been read back. */
typedef int (*toyni_fri_challenge_fn)(void* user, unsigned round, const uint8_t* prev_root32, uint32_t* beta_out);
int toyni_fri_commit_phase_device(toyni_ntt_ctx* ctx, const uint32_t* d_layer0, size_t m0, uint32_t x0, size_t final_size,
                                  const uint8_t* d_salts, toyni_fri_challenge_fn challenge, void* user, uint32_t* d_layers,
                                  uint8_t* d_levels, uint8_t* h_roots, unsigned* rounds_out, void* stream);
/* Constraint and quotient evaluations (src/fibonacci.rs:133-150): with n = N >> log_blowup, g = w_n, T(g x_i) = trace[(i + B) mod N]:
 *   c_i = (T(g^2 x_i) - (T(g x_i) + T(x_i))) (x_i - g^(n-1)) (x_i - g^(n-2)),   q_i = c_i / (x_i^n - 1).
 * d_c_evals may be NULL.  TOYNI_E_ZERO_INVERSE if Z_H vanishes on the coset (shift^n a B-th root of unity). */
int toyni_fib_quotient_device(toyni_ntt_ctx* ctx, const uint32_t* d_trace_lde, uint32_t* d_c_evals, uint32_t* d_q_evals, unsigned log_blowup,
                              uint32_t shift, void* stream);
/* DEEP layer (src/fibonacci.rs:186-198): d_i = ((q_i - q_z) + (T(g^2 x_i) - t_ggz) + (T(g x_i) - t_gz) + (T(x_i) - t_z)) / (x_i - z);
 * ood = {t_z, t_gz, t_ggz, q_z}.  z must lie outside the coset (derive_z_from_transcript, :379-399, guarantees it); a point
 * with x_i = z yields 0 for that point alone (the reference panics: "Cannot invert zero"). */
int toyni_fib_deep_device(toyni_ntt_ctx* ctx, const uint32_t* d_trace_lde, const uint32_t* d_q_evals, uint32_t* d_out, unsigned log_blowup,
                          uint32_t shift, uint32_t z, const uint32_t ood[4], void* stream);
/* Polynomial::evaluate (src/math/polynomial.rs:134-144) of ncoeffs device-resident coefficients at 1..4 points (host values):
 * d_out[p] = sum_i c_i points[p]^i.  The OOD evaluations t_z, t_gz, t_ggz share one read of the coefficients. */
int toyni_poly_eval_device(toyni_ntt_ctx* ctx, const uint32_t* d_coeffs, size_t ncoeffs, const uint32_t* points, unsigned npoints,
                           uint32_t* d_out, void* stream);
/* Openings (open_merkle, src/fibonacci.rs:366-375, over MerkleTree::get_proof, src/merkle.rs:50-80) of nidx leaves of a tree built
 * by toyni_merkle_commit_device / toyni_fri_fold_commit_device, gathered on the device into nidx records of
 * toyni_merkle_open_record_bytes(n) bytes each:  depth x 32 path bytes | 16 salt bytes (zero if d_salts is NULL) | the value as
 * 8 LE bytes | depth position bytes (1 = the sibling is the LEFT input), padded to a multiple of 8.  d_out 8-byte aligned. */
size_t toyni_merkle_open_record_bytes(size_t n);
int toyni_merkle_open_device(const uint8_t* d_levels, size_t n, const uint32_t* d_values, const uint8_t* d_salts, const uint32_t* d_indices,
                             size_t nidx, uint8_t* d_out, void* stream);
/* The same for several trees at once (the query phase of a proof opens the trace, quotient and DEEP trees and every FRI layer's:
 * src/fibonacci.rs:249-295): one launch per 32 trees instead of one per tree.  Every field as in toyni_merkle_open_device. */
typedef struct {
    const uint8_t* d_levels;
    size_t n;
    const uint32_t* d_values;
    const uint8_t* d_salts;
    const uint32_t* d_indices;
    size_t nidx;
    uint8_t* d_out;
} toyni_merkle_open_group;
int toyni_merkle_open_groups_device(const toyni_merkle_open_group* groups, size_t ngroups, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 4. Plumbing
 * ---------------------------------------------------------------------------------------------- */
int toyni_malloc(void** d_ptr, size_t bytes);
/* Pinned (page-locked) host memory.  A host-slice entry point that is handed pinned memory (from here, or registered by the
 * caller with hipHostRegister) and at least two chunks of data (2 x 64 MiB; TOYNI_PIPE_CHUNK_BYTES) runs pipelined: the upload of
 * chunk k + 1, the kernels of chunk k and the download of chunk k - 1 overlap on three streams (PCIe is full duplex).  Pageable
 * memory (a Rust Vec, what the reference passes: src/ntt.rs:233) takes the plain upload - kernels - download path: its copies
 * are staged by the runtime and do not overlap. */
int toyni_host_alloc(void** h_ptr, size_t bytes);
int toyni_host_free(void* h_ptr);
int toyni_free(void* d_ptr);
int toyni_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int toyni_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
/* Stream-ordered copies and fills (asynchronous with pinned host memory; pageable host memory makes them block like HIP does). */
int toyni_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream);
int toyni_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream);
int toyni_memcpy_d2d_async(void* d_dst, const void* d_src, size_t bytes, void* stream);
int toyni_memset_async(void* d_ptr, int value, size_t bytes, void* stream);
/* Merkle salts on the device: bytes of the ChaCha20 keystream (RFC 8439: 256-bit key, block counter from 0, 96-bit nonce =
 * 0 || nonce as two little-endian words).  The reference draws 16 bytes per leaf from rand::thread_rng(), a ChaCha CSPRNG on the
 * host (src/fibonacci.rs:341-343); this keeps ~130 MB of salts per 2^16-row proof off PCIe.  bytes: a multiple of 64; d_out
 * 16-byte aligned.  The key is the caller's secret: draw it from the OS (getrandom) per proof. */
int toyni_chacha20_fill_device(void* d_out, size_t bytes, const uint8_t key[32], uint64_t nonce, void* stream);
int toyni_narrow_u64_to_u32(const uint64_t* d_in, uint32_t* d_out, size_t count, void* stream);
int toyni_widen_u32_to_u64(const uint32_t* d_in, uint64_t* d_out, size_t count, void* stream);
/* Streams for host languages that bind only this library: device -1 = the current device; the stream is non-blocking with respect
 * to HIP's legacy default stream.  toyni_stream_destroy waits for the stream's work first. */
int toyni_stream_create(void** stream, int device);
int toyni_stream_destroy(void* stream);
/* Ordering between two streams: what is enqueued on `stream` after this call runs after everything enqueued on `after` so far. */
int toyni_stream_wait(void* stream, void* after);
/* The same in two steps: mark a point of one stream now (toyni_event_record), make another stream wait for it later. */
int toyni_event_create(void** event);
int toyni_event_destroy(void* event);
int toyni_event_record(void* event, void* stream);
int toyni_stream_wait_event(void* stream, void* event);
int toyni_stream_synchronize(toyni_ntt_ctx* ctx, void* stream);   /* hipStreamSynchronize(stream); with a context: also releases what that stream outgrew */
int toyni_ntt_ctx_trim(toyni_ntt_ctx* ctx);                       /* hipDeviceSynchronize, then frees every intermediate buffer of the context */
/* Stream capture: a warm context only enqueues kernels, so its device-resident calls can be captured into a HIP graph.  One caveat:
 * after a context has outgrown or evicted an intermediate buffer, the next call on it that finds the buffer's event complete frees it
 * (hipFree: a device-wide synchronisation, illegal under a GLOBAL-mode capture).  The library skips that sweep while a stream the
 * CURRENT call enqueued on (or the context's own stream) is capturing, and runs it in relaxed mode on the calling thread; it never asks
 * about streams of earlier calls (the caller may have destroyed them since), so a global-mode capture that another thread has open on
 * some other stream is not visible to it.  Callers that capture on several threads use hipStreamCaptureModeThreadLocal / Relaxed, or
 * call toyni_ntt_ctx_trim (or a blocking entry point) before they start capturing.  A stream the context has carried may be destroyed
 * at any time; calling toyni_stream_synchronize(ctx, stream) first also releases what the context kept for it. */
int toyni_set_device(int device);
/* Diagnostics: the symbols (one per line) of every kernel this process has launched through the library so far.  Returns the bytes
 * needed including the terminating 0; (NULL, 0) asks for the size.  An in-memory list only: the library writes no file and reads no
 * environment variable for it.  Used by the test suite to prove that every kernel of the shipped binary was run (child processes of a
 * test session dump this list at exit through the test harness). */
size_t toyni_launched_kernels(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TOYNI_HIP_H */
