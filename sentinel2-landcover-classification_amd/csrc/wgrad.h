// Shared declarations of the weight-gradient kernels (wgrad.hip: generic 4-wave kernels; wgrad_pc.hip: the
// producer / consumer kernels for the MFMA-bound shapes).
#pragma once
#include "common.h"

namespace s2k {

constexpr int WG_EPT_MAX = 2;  // halo elements per thread per channel (tile <= 512 floats / channel)

enum { WG_PIX = 0, WG_SPATIAL = 1, WG_GATHER = 2 };

struct WgradP {
    const float *p, *bnvp, *gatep, *q, *bnvq, *gateq;
    float* wgs;
    int B, M, C, CTOT, H, W, KH, KW, S, PT, PL, HO, WO, prop, proq;
    int T, n_mtiles, n_ctiles, HWp, HWq, ntiles, tiles_per_split;
    int NP;                      // pixels per chunk (PIX / GATHER)
    int R, XW, XWe, tiles_x, tiles_y, IR, IC, WS, CSQ, PSTR;
    int p_bf16;                  // P is stored as bf16 (opdefs WGRAD.P_BF16; only wgrad_bf16.hip's 1x1 kernel reads such a tensor)
    int streamk;                 // wgrad_q4.hip: equal ranges of the (output tile, pixel tile) list per workgroup instead of pixel splits of one tile
    int exp;                     // tuning builds only (S2K_WG_EXP): 1 = stage the first tile only, 2 = no combine, 4 = no MFMA loop
};

// wgrad_pc.hip: returns S2K_OK when it launched the stage, 1 when the shape is not one of its instantiations (the caller
// then takes the generic kernel), or a negative error code
int launch_wgrad_pc(WgradP& p, int mode, hipStream_t st);
// wgrad_q4.hip (1x1 / Linear weight gradients with H * W % 4 == 0: quad operand reads): same return convention
int launch_wgrad_q4(WgradP& p, int mode, hipStream_t st);
// wgrad_bf16.hip (stages carrying S2K_FLAG_BF16 in bf16-mixed plans): same return convention
int launch_wgrad_bf16(WgradP& p, int mode, hipStream_t st);

// bijective remap: consecutive logical tiles land on the same XCD (hardware deals blocks round-robin over the 8 XCDs;
// which XCD is irrelevant, only that ids congruent mod 8 share one)
__device__ __forceinline__ int wg_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Combine of a workgroup's accumulator tiles: wgs[t][m][c] += acc.
// Atomics on one address execute one after the other at the memory side, and a thin layer's whole gradient is a few hundred
// addresses that EVERY workgroup adds to: on the 24 x 24 project conv of block 0 (512 pixel splits x 4 k-waves) the combine took
// 74 of the bf16 kernel's 92 us.  So the WVK k-waves of an accumulator tile first add their partial sums through LDS (`red`:
// (WVK - 1) * (4 / WVK) * 1024 floats, free once every wave has left the multiply loop - the caller's barrier), and one wave per
// tile issues the atomics.  Every wave of the (consumer) group must call this; `nsync` = number of threads at the barriers
// (all of them: __syncthreads).
template <int T, int WM, int WN, int WVK>
__device__ __forceinline__ void wg_combine(const WgradP& p, f32x16 (&acc)[T][WM][WN], float* red, int wk, int wmn, int lane,
                                           int m0, int c0, int wm0, int wc0) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) {
                if constexpr (WVK > 1) {
                    if (wk != 0) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) red[((wmn * (WVK - 1) + wk - 1) * 16 + reg) * 64 + lane] = acc[t][rm][rn][reg];
                    }
                    __syncthreads();
                    if (wk == 0) {
#pragma unroll
                        for (int k = 0; k < WVK - 1; ++k)
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) acc[t][rm][rn][reg] += red[((wmn * (WVK - 1) + k) * 16 + reg) * 64 + lane];
                    }
                }
                if (WVK == 1 || wk == 0) {
                    const int gc = c0 + wc0 + rn * 32 + l31;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        if (gm < p.M && gc < p.C)
                            atomicAdd(p.wgs + ((int64_t)t * p.M + gm) * p.CTOT + gc, acc[t][rm][rn][reg]);
                    }
                }
                if constexpr (WVK > 1) __syncthreads();       // `red` is rewritten for the next accumulator tile
            }
}

}  // namespace s2k
