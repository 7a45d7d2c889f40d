// Shared device/host helpers for the s2k kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "../../include/s2k.h"

namespace s2k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int NTHREADS = 256;

// ---- prologue: v' = act(scale*v + shift) (* gate) -------------------------------------------
// v_exp_f32 + v_rcp_f32 (~1 ulp each): far inside the 1e-3 logits bar (measured 2e-5) and 3.6 % faster end to end than
// expf + a true division.  Cost: the median fp32 gradient noise of a b0 train step (train-mode BatchNorm on 7x7 maps
// amplifies every ulp) is 2.6x the fp32 oracle's own distance from float64 instead of 2.0x (tests/test_unet_gpu.py bars: 3x).
// -DS2K_EXACT_SILU (tools/exp_exact_silu.sh builds libs2k_exact.so with it; never the shipped library): expf + a true division, so
// that what the fast form costs in parity terms is a measured table (profiles/r04_exact_silu.md), not a comment.
#ifdef S2K_EXACT_SILU
__device__ __forceinline__ float sigmoid_act(float u) { return 1.0f / (1.0f + expf(-u)); }
#else
__device__ __forceinline__ float sigmoid_act(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
#endif
__device__ __forceinline__ float silu_f(float u) { return u * sigmoid_act(u); }

// exact (erf) GELU, as nn.GELU() / F.gelu default (timm Mlp, prithvi_segmentation.py:58,62)
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }

__device__ __forceinline__ float apply_pro(float v, int pro, float scale, float shift) {
    if (pro != S2K_PRO_NONE) {
        v = fmaf(v, scale, shift);
        if (pro == S2K_PRO_SILU) v = silu_f(v);
        else if (pro == S2K_PRO_RELU) v = fmaxf(v, 0.0f);
        else if (pro == S2K_PRO_GELU) v = gelu_f(v);
    }
    return v;
}

// compile-time activation: lets a kernel hoist the (launch-uniform) prologue kind out of its unrolled
// element loops instead of branching per element
template <int PRO>
__device__ __forceinline__ float apply_pro_c(float v, float scale, float shift) {
    if (PRO == S2K_PRO_NONE) return v;
    const float u = fmaf(v, scale, shift);
    if (PRO == S2K_PRO_SILU) return silu_f(u);
    if (PRO == S2K_PRO_RELU) return fmaxf(u, 0.0f);
    if (PRO == S2K_PRO_GELU) return gelu_f(u);
    return u;
}
// call f(std::integral_constant<int, pro>) for the runtime value `pro`
template <typename F>
__device__ __forceinline__ void dispatch_pro(int pro, F&& f) {
    if (pro == S2K_PRO_NONE) f(std::integral_constant<int, S2K_PRO_NONE>{});
    else if (pro == S2K_PRO_SILU) f(std::integral_constant<int, S2K_PRO_SILU>{});
    else if (pro == S2K_PRO_RELU) f(std::integral_constant<int, S2K_PRO_RELU>{});
    else if (pro == S2K_PRO_GELU) f(std::integral_constant<int, S2K_PRO_GELU>{});
    else f(std::integral_constant<int, S2K_PRO_AFFINE>{});
}

// d act(u) / du for act in {none, silu, relu}
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == S2K_PRO_SILU) {
        float s = sigmoid_act(u);
        return s * (1.0f + u * (1.0f - s));
    }
    if (act == S2K_PRO_RELU) return u > 0.0f ? 1.0f : 0.0f;
    if (act == S2K_PRO_GELU) return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * __expf(-0.5f * u * u) * 0.39894228040143268f;
    return 1.0f;
}

// CONV epilogue with RES: v + r, or v * act'(r) when the stage carries S2K_FLAG_RES_GELU_GRAD (res_mul = the activation; kernel-uniform)
__device__ __forceinline__ float res_combine(float v, float r, int res_mul) { return res_mul ? v * act_grad(r, res_mul) : v + r; }

// ---- bounds-checked buffer loads ----------------------------------------------------------------------
// Operand tiles that hang over a tensor edge are fetched through a buffer descriptor sized to the tensor:
// the hardware returns 0 for any offset beyond it, so the stagers keep ONE affine address stream (base +
// i*stride, 32-bit offsets) with no per-row clamping and no branch around any load.
// IMPORTANT: invalid elements are expressed as an out-of-range OFFSET (BUF_OOB), never as
// `cond ? load : 0` — hipcc sinks a load under a select into a branch and then waits for each one.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t BUF_OOB = 0x7ffffff0u;   // >= any descriptor size we create (tensors are < 2 GiB, checked on the host)
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes), 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// vector offset (range-checked: BUF_OOB -> 0) + scalar offset (NOT range-checked: keep base + soff inside the tensor)
__device__ __forceinline__ float bload_s(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 bload4(rsrc_t r, uint32_t byte_off) {   // 16-byte aligned offsets only
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ uint32_t bload_u16(rsrc_t r, uint32_t byte_off) {      // one 16-bit element, zero-extended (out of range -> 0)
    return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0);
}
// (the vector the b64 builtin returns must be bit-cast as a WHOLE: indexing it - v[0], v[1] - makes this compiler emit a single
// buffer_load_dword and copy the dword into both elements)
__device__ __forceinline__ float2 bload2(rsrc_t r, uint32_t byte_off) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
    return make_float2(__builtin_bit_cast(float, (uint32_t)(u & 0xffffffffull)), __builtin_bit_cast(float, (uint32_t)(u >> 32)));
}

// ---- reductions -------------------------------------------------------------------------------
// Cross-lane sums run on the DPP data path of the vector ALU (one v_add_f32_dpp per step); the generic
// __shfl_xor lowers to ds_bpermute_b32 = an LDS round trip + s_waitcnt per step, which made the BatchNorm
// statistics of a short-K conv cost more than its MFMAs.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov0(float v) {   // lanes outside ROW_MASK (or without a source lane) read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <- lane 7-i of its 8-lane half
constexpr int DPP_MIRROR = 0x140;       // lane i <- lane 15-i of its 16-lane row
constexpr int DPP_BCAST15 = 0x142;      // lane 15 of every row -> all lanes of the next row
constexpr int DPP_BCAST31 = 0x143;      // lane 31 -> all lanes of rows 2 and 3
// every lane ends with the sum of its 16-lane row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov0<DPP_XOR1>(v);
    v += dpp_mov0<DPP_XOR2>(v);
    v += dpp_mov0<DPP_HALF_MIRROR>(v);
    v += dpp_mov0<DPP_MIRROR>(v);
    return v;
}
// sum over the 32 lanes that share (lane >> 5); valid in lanes 16..31 (resp. 48..63) ONLY
__device__ __forceinline__ float half_sum_hi(float v) {
    v = row16_sum(v);
    v += dpp_mov0<DPP_BCAST15, 0xa>(v);
    return v;
}
// sum over the wave; valid in lanes 48..63 ONLY
__device__ __forceinline__ float wave_sum_hi(float v) {
    v = half_sum_hi(v);
    v += dpp_mov0<DPP_BCAST31, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {   // all lanes (the total comes back through an SGPR)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_hi(v)), 63));
}
// sum over aligned groups of lpp lanes (lpp = wave-uniform power of two), all lanes of the group
__device__ __forceinline__ float group_sum(float v, int lpp) {
    if (lpp >= 2) v += dpp_mov0<DPP_XOR1>(v);
    if (lpp >= 4) v += dpp_mov0<DPP_XOR2>(v);
    if (lpp >= 8) v += dpp_mov0<DPP_HALF_MIRROR>(v);
    if (lpp >= 16) v += dpp_mov0<DPP_MIRROR>(v);
    if (lpp >= 32) v += __shfl_xor(v, 16, 64);
    if (lpp >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 32 lanes that share (lane >> 5), all lanes
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of a double; result valid in thread 0.  `red` = shared scratch of >= 4 doubles.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += red[i];
    }
    return r;
}

__device__ __forceinline__ void atomic_add_d(double* p, double v) { atomicAdd(p, v); }

// ---- BN_FINALIZE folded into the first consumer of {scale, shift} (plan/opdefs.py FOLD_*) -------------------------------
// Every workgroup (wave, thread) that needs a channel's {scale, shift} derives them from the statistics replicas itself -
// the same arithmetic everywhere, so all of them agree bit for bit - and exactly one of them per channel (`writer`) stores
// BNV = {scale, shift, mean, invstd}[C] for the later readers (the backward) and applies the running-statistics update.
struct BnFold {
    const double* stats;      // [nrep][2][C] sum / sum of squares; nullptr = not folded (BNV is read)
    const float *gamma, *beta;
    float *rm, *rv, *bnv;
    double inv_count, unbias;  // 1 / count, count / max(count - 1, 1)
    float eps, mom;
    int nrep;
};

__device__ __forceinline__ void bn_fold_finish(const BnFold& f, int C, int c, double s, double q, bool writer, float& scale, float& shift) {
    // mean and variance in f64 (the subtraction cancels).  This runs once per WAVE of the consumer (B times per channel), where
    // an f64 divide + square root cost more than the launch they replace: the reciprocal square root is an f32 estimate plus
    // one Newton step in f64 (relative error ~1e-14 before the rounding to f32, i.e. the value BN_FINALIZE's 1 / sqrt gives)
    const double m = s * f.inv_count;
    double var = fma(q, f.inv_count, -m * m);
    if (var < 0.0) var = 0.0;
    const double xx = var + (double)f.eps;
    double yy = (double)rsqrtf((float)xx);
    yy = yy * (1.5 - 0.5 * xx * yy * yy);
    const float invstd = (float)yy;
    const float mean = (float)m;
    scale = f.gamma[c] * invstd;
    shift = f.beta[c] - mean * scale;
    if (writer) {
        const double unbiased = var * f.unbias;
        f.rm[c] = (1.0f - f.mom) * f.rm[c] + f.mom * mean;
        f.rv[c] = (1.0f - f.mom) * f.rv[c] + f.mom * (float)unbiased;
        f.bnv[c] = scale;
        f.bnv[C + c] = shift;
        f.bnv[2 * C + c] = mean;
        f.bnv[3 * C + c] = invstd;
    }
}
// one thread per channel (replicas added in order)
__device__ __forceinline__ void bn_fold_thread(const BnFold& f, int C, int c, bool writer, float& scale, float& shift) {
    double s = 0.0, q = 0.0;
    for (int r = 0; r < f.nrep; ++r) { s += f.stats[(int64_t)r * 2 * C + c]; q += f.stats[(int64_t)r * 2 * C + C + c]; }
    bn_fold_finish(f, C, c, s, q, writer, scale, shift);
}
// one converged wave per channel (c wave-uniform); the result is in every lane, `writer` is honoured by lane 0
__device__ __forceinline__ void bn_fold_wave(const BnFold& f, int C, int c, bool writer, float& scale, float& shift) {
    const int lane = threadIdx.x & 63;
    c = __builtin_amdgcn_readfirstlane(c);         // tell the compiler: scalar loads below
    double s = 0.0, q = 0.0;
    if (f.nrep <= 8) {       // wave-uniform addresses: scalar loads
        for (int r = 0; r < f.nrep; ++r) { s += f.stats[(int64_t)r * 2 * C + c]; q += f.stats[(int64_t)r * 2 * C + C + c]; }
    } else {
        for (int r = lane; r < f.nrep; r += 64) { s += f.stats[(int64_t)r * 2 * C + c]; q += f.stats[(int64_t)r * 2 * C + C + c]; }
        s = wave_sum_d(s);
        q = wave_sum_d(q);
    }
    bn_fold_finish(f, C, c, s, q, writer && lane == 0, scale, shift);
}

// ---- host side --------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

// Tuning / A-B switches exist only in a build made with -DS2K_TUNING (tools/, never the shipped libs2k.so): in the product
// library every switch is its compiled-in default, so a leaked environment variable cannot change kernels or results.
inline int tune_int(const char* name, int dflt) {
#ifdef S2K_TUNING
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Which kernel family a stage's launcher picked (read by s2k_program_profile_variants): 0 = the stage's generic kernel,
// 1 = the producer/consumer kernel (conv_pc_kernel / wgrad_pc_kernel).
extern thread_local int g_s2k_variant;

// hipFuncSetAttribute acts on the current device: run a kernel's attribute call once per device.  std::call_once makes every
// other host thread WAIT until the first caller's attribute call has returned (include/s2k.h allows concurrent calls: a second
// thread must not launch a > 64 KB dynamic-LDS kernel before the attribute has landed).
struct PerDeviceOnce {
    std::once_flag flags[64];
    template <typename F>
    void run(F&& f) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { f(); return; }
        std::call_once(flags[d], f);
    }
};

struct Ctx {
    void* const* bases;
    int n_bases;
    hipStream_t stream;
};

// host side of BnFold: the five tensor refs FSTATS, FGAMMA, FBETA, FRM, FRV (consecutive t slots) + scalars of a stage record
inline int fill_bn_fold(BnFold& f, const Ctx& c, const int64_t* t5, int64_t count, int nrep, float eps, float mom, float* bnv,
                        const char* who) {
    f = BnFold{};
    if (t5[0] < 0) return S2K_OK;                       // not folded
    void* ptr[5];
    for (int i = 0; i < 5; ++i) {
        const int64_t ref = t5[i];
        const int base = (int)(ref >> 56);
        if (ref < 0 || base >= c.n_bases || c.bases[base] == nullptr) { set_error("%s: folded BN_FINALIZE needs FSTATS, FGAMMA, FBETA, FRM, FRV", who); return S2K_EINVAL; }
        ptr[i] = static_cast<char*>(c.bases[base]) + (ref & ((1ll << 56) - 1));
    }
    if (count <= 0 || !bnv || bnv == reinterpret_cast<float*>(1)) { set_error("%s: folded BN_FINALIZE needs FCOUNT > 0 and BNV", who); return S2K_EINVAL; }
    f.stats = static_cast<const double*>(ptr[0]);
    f.gamma = static_cast<const float*>(ptr[1]);
    f.beta = static_cast<const float*>(ptr[2]);
    f.rm = static_cast<float*>(ptr[3]);
    f.rv = static_cast<float*>(ptr[4]);
    f.bnv = bnv;
    f.inv_count = 1.0 / (double)count;
    f.unbias = (double)count / (count > 1 ? (double)(count - 1) : 1.0);
    f.eps = eps;
    f.mom = mom;
    f.nrep = nrep > 0 ? nrep : 1;
    return S2K_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace s2k
