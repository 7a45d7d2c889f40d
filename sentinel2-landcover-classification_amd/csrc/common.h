// Shared device/host helpers for the s2k kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s2k.h"

namespace s2k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int NTHREADS = 256;

// ---- prologue: v' = act(scale*v + shift) (* gate) -------------------------------------------
__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }

__device__ __forceinline__ float apply_pro(float v, int pro, float scale, float shift) {
    if (pro != S2K_PRO_NONE) {
        v = fmaf(v, scale, shift);
        if (pro == S2K_PRO_SILU) v = silu_f(v);
        else if (pro == S2K_PRO_RELU) v = fmaxf(v, 0.0f);
    }
    return v;
}

// d act(u) / du for act in {none, silu, relu}
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == S2K_PRO_SILU) {
        float s = 1.0f / (1.0f + __expf(-u));
        return s * (1.0f + u * (1.0f - s));
    }
    if (act == S2K_PRO_RELU) return u > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}

// ---- reductions -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 32 lanes that share (lane >> 5)
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

// ---- host side --------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

struct Ctx {
    void* const* bases;
    int n_bases;
    hipStream_t stream;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace s2k
