/* s2k — C ABI of the MI355X-native segmentation hot path (libs2k.so, gfx950 only).
 *
 * Drop-in boundary.  The reference's hot path is pure Python calling ATen:
 *   EfficientnetUnet.forward            /root/reference/src/modules/efficientnet_unet.py:125-138
 *   EfficientNet.encode / MBConvBlock   efficientnet_unet.py:251-263, :377-387
 *   FocalLoss.__call__ / CrossEntropy   /root/reference/src/losses.py:24-89
 *   logits.argmax(dim=1)                /root/reference/src/train_segmentation.py:145
 *   loss.backward() (torch autograd)    train_segmentation.py:87-93 via Lightning
 * There is no FFI in the reference for this path, so the entry points below are what a binding
 * for it would call: a whole forward (or backward) of the network is ONE `s2k_program_run` over
 * an array of fused-stage records planned on the host; every stage kind is also launchable on
 * its own (`s2k_op_launch`) for parity tests.  INTEGRATION.md shows the ctypes stub.
 *
 * Rules of the boundary
 *   - plain C: pointers, sizes, POD structs; no torch / HIP types in signatures except the
 *     stream, passed as an opaque `void*` (a hipStream_t);
 *   - every device pointer is BORROWED: the library never allocates, frees or synchronises;
 *     all work is enqueued on the caller's stream (graph-capturable);
 *   - return 0 on success, a negative S2K_E* code on failure, never throw; the message for the
 *     last failure on the calling thread is `s2k_last_error()`;
 *   - re-entrant: may be called concurrently from several host threads, on several streams and
 *     on several devices of one process.  The device that owns `stream` is made current for the
 *     duration of a call and restored afterwards (a NULL stream means the caller's current
 *     device).  The only mutable global is a mutex-guarded pool of per-device side streams;
 *     a side stream + event pair is leased for ONE s2k_program_run and returned when it ends,
 *     by which time all of its work has been ordered back onto the caller's stream;
 *   - no environment variable changes what the shipped library computes (tuning switches exist
 *     only in builds made with -DS2K_TUNING).
 */
#ifndef S2K_H
#define S2K_H

#include <stddef.h>
#include <stdint.h>

#include "s2k_ops.h"

#ifdef __cplusplus
extern "C" {
#endif

#define S2K_ABI_VERSION 2

#define S2K_OK 0
#define S2K_EINVAL (-22)   /* malformed stage record / unsupported geometry */
#define S2K_ENOSYS (-38)   /* unknown stage kind */
#define S2K_EFAULT (-14)   /* null base for a referenced tensor */
#define S2K_EHIP (-5)      /* a HIP launch failed; see s2k_last_error() */

/* One fused stage.  Field meaning per kind: include/s2k_ops.h (generated from plan/opdefs.py).
 * t[]: tensor refs = (base_id << 56) | byte_offset, -1 = null.  256 bytes. */
typedef struct S2kOp {
    int32_t kind;
    int32_t flags;
    int64_t t[S2K_N_T];
    int64_t n[S2K_N_N];
    int32_t d[S2K_N_D];
    float f[S2K_N_F];
} S2kOp;

int s2k_abi_version(void);
size_t s2k_op_size(void);                 /* sizeof(S2kOp), for binding self-checks */
const char* s2k_last_error(void);         /* thread-local, never NULL */
const char* s2k_kind_name(int kind);      /* "CONV", ... or NULL */

/* Enqueue ops[begin, end) on `stream`.  bases[i] is the device pointer of base i
 * (S2K_BASE_*), NULL if that base is not used by the range. */
int s2k_program_run(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream);

/* Single stage (parity tests, loss, argmax). */
int s2k_op_launch(const S2kOp* op, void* const* bases, int n_bases, void* stream);

/* Device-side timing of a program range with HIP events on `stream` (bench.py roofline leg):
 * per-kind accumulated milliseconds into ms_by_kind[S2K_N_KINDS + 1]; synchronises the stream. */
int s2k_program_profile(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream,
                        float* ms_by_kind, int* launches_by_kind);

/* same, one entry per stage: ms_per_op[end - begin] */
int s2k_program_profile_ops(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream,
                            float* ms_per_op);

/* same, plus which kernel the stage's launcher picked: variant_per_op[i] = 0 for the stage's generic kernel
 * (conv_igemm_kernel, wgrad_kernel, ...), 1 for the producer/consumer kernel (conv_pc_kernel / wgrad_pc_kernel), 2 for the
 * bf16 MFMA kernels (conv_bf16_kernel / wgrad_bf16_kernel; S2K_FLAG_BF16 stages), 3 for the LDS-DMA ring kernel
 * (conv_dma_kernel), 4 for the quad-read kernels (conv_q4_kernel: S2K_FLAG_Q4 stages; wgrad_q4_kernel: picked by the launcher).
 * bench.py uses it to attribute time and algorithmic FLOPs to the kernel names a rocprofv3 trace shows. */
int s2k_program_profile_variants(const S2kOp* ops, int begin, int end, void* const* bases, int n_bases, void* stream,
                                 float* ms_per_op, int* variant_per_op);

/* fused Adam step (L2-coupled decay, train_segmentation.py:109-127): p, g, m, v flat fp32 [n].  Hyper-parameters are
 * doubles, as Python hands them to torch.optim.Adam: `1 - beta2` and `lr / (1 - beta1**step)` are formed in double and
 * rounded to fp32 once, which is what makes the update bit-compatible with torch's (ABI 2; ABI 1 took floats). */
int s2k_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, void* stream);

/* Measured ceilings of the current device for roofline reporting (bench.py): sustained v_mfma_f32_32x32x2_f32 rate with every
 * SIMD issuing back to back (TFLOP/s), the shader clock held meanwhile (MHz), and a float4 stream copy (GB/s, read + write).
 * `scratch`: >= 64 MiB of device memory (size the copy past the caches: 1 GiB+).  Synchronises `stream`.  Not on the hot path. */
int s2k_measure_peaks(void* scratch, size_t scratch_bytes, int waves_per_simd, double* mfma_tflops, double* mfma_clock_mhz,
                      double* copy_gbps, void* stream);

/* Diagnostics for tests: run a 32x32x2 f32 MFMA on A[32x2], B[2x32] and return D[32x32]
 * (checks the lane maps the kernels rely on with exact integer data). */
int s2k_selftest_mfma(const float* a, const float* b, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S2K_H */
