/*
 * toyni_hip_tools.h -- measurement hooks of the MEASUREMENT BUILD libtoyni_hip_tools.so (the same translation unit as
 * libtoyni_hip.so compiled with -DTOYNI_TOOLS).  They are NOT part of the shipped library or of the drop-in boundary
 * (include/toyni_hip.h): bench.py uses them to price the pass kernels with HIP events on the launch stream (roofline object).
 * The instruction-rate probe is its own program, tools/microbench.hip.
 */
#ifndef TOYNI_HIP_TOOLS_H
#define TOYNI_HIP_TOOLS_H

#include "toyni_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-pass kernel timing for bench.py's roofline object: launches pass p of `batch` transforms `reps` times
 * between HIP events on `stream` (pass 0 reads d_data, later passes the context's work buffer; values stay
 * canonical but d_data's contents are overwritten).  ms_per_pass[p] = average launch duration in ms for
 * p < toyni_ntt_ctx_passes(ctx).  Blocking. */
int toyni_ntt_profile_passes(toyni_ntt_ctx* ctx, uint32_t* d_data, size_t batch, int inverse, int reps, float* ms_per_pass, void* stream);

/* Launch durations of the pass kernels INSIDE a workload: while enabled, every pass launch this context enqueues
 * (toyni_ntt_device and friends) is bracketed by a pair of HIP events on the launch stream.  _read waits for the
 * recorded events, adds the durations into ms_sum[direction * 3 + pass] / launches[direction * 3 + pass]
 * (direction 0 = forward, 1 = inverse; both arrays hold 6 entries, zero-filled first) and drops the records.
 * bench.py brackets its timed region with enable / read, so the roofline object prices the launches that were timed. */
int toyni_ntt_ctx_timing(toyni_ntt_ctx* ctx, int enable);
int toyni_ntt_ctx_timing_read(toyni_ntt_ctx* ctx, float* ms_sum, uint32_t* launches);

/* Fault injection into the multi-device forms (tests only; csrc/multi_gpu.hpp): bit 0 = every pair of different lanes is treated
 * as two devices without peer access (group creation must fail with TOYNI_E_NO_PEER_ACCESS); bit 1 = the first-use self-check
 * also runs for groups whose lanes all sit on one device; bit 2 = that self-check sees one corrupted word (must fail with
 * TOYNI_E_SELF_CHECK).  0 restores normal behaviour. */
int toyni_tools_inject(unsigned flags);
/* NCCL_VERSION_CODE reported by the librccl that dlopen found (0: not loadable / no ncclGetVersion). */
int toyni_tools_rccl_version(void);
/* Where a step of the single-process multi-device transform (toyni_ntt_slab_multi_gpu_*) spends its time: while enabled, lane 0's
 * compute stream carries a HIP event between the four stages of every transform (one-piece exchanges only).  _read waits for them and
 * adds up ms[0..3] = slab pass / exchange (the wait for the incoming blocks) / relayout / row transforms over *transforms transforms. */
int toyni_tools_slab_phases(int enable);
int toyni_tools_slab_phases_read(float* ms, unsigned* transforms);

#ifdef __cplusplus
}
#endif
#endif /* TOYNI_HIP_TOOLS_H */
