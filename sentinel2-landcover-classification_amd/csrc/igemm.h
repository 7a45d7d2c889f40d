// Shared declarations of the implicit-GEMM convolution kernels (igemm.hip: generic 4-wave kernels; igemm_pc.hip: the
// producer / consumer kernels for the prologue-light, MFMA-bound shapes).
#pragma once
#include "common.h"

namespace s2k {

enum { BM_PIX = 0, BM_SPATIAL = 1 };

struct ConvP {
    const float *x1, *bnv1, *gate1, *x2, *bnv2, *wt, *bias, *res;
    int x1_bf16;         // X1 is stored as bf16 (opdefs CONV.X1_BF16; only conv_bf16.hip's 1x1 kernel reads such a tensor)
    int tepi;            // conv_igemm_kernel: 1 = transposed (LDS) store of the output tile, set by its launcher
    int res_mul;         // S2K_FLAG_RES_GELU_GRAD: the epilogue multiplies by act'(RES) (= S2K_PRO_GELU) instead of adding RES; else 0
    int force_dma;       // S2K_FLAG_DMA: the LDS-DMA ring kernel for every shape it supports (tests), not only where its routing rule sends a stage
    int exp;             // tuning builds only (S2K_CV_EXP): 1 = no epilogue (nothing stored), 2 = no MFMA loop
    const void* wtb;     // bf16 copy of the packed weights ([KP/8][T][MP][8], WEIGHT_PACK BF16_BASE) when the stage carries S2K_FLAG_BF16, else null
    const float* wtq;    // f32 quad copy of the packed weights ([KP/8][MP][8], WEIGHT_PACK Q4_BASE) when the stage carries S2K_FLAG_Q4, else null
    float* y;
    float* scratch;      // split-K partial tiles [splits][Y layout] (deep, short-N layers), or null
    int splits, chunks_per_split;
    int64_t y_elems;
    double* stats;
    int B, C1, C2, H, W, M, KH, KW, S, PT, PL, HO, WO;
    int pro1, pro2, mode, w_sm, w_sk, w_st, flip, beta, YC, nrep;
    int Ctot, n_mtiles, n_tiles, HW, Ntot, a_mfast, b_floats;
    int R, XW, tiles_x, tiles_y, IR, IC, WS, CS;
};

// bijective remap: consecutive logical tiles land on the same XCD (hardware deals blocks round-robin
// over the 8 XCDs; which XCD is irrelevant, only that ids congruent mod 8 share one)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// igemm.hip: the split-K tail - Y (+)= sum of the partial tiles in p.scratch (fixed order) + bias + residual, BatchNorm statistics
void launch_splitk_reduce(const ConvP& p, hipStream_t st);

// igemm_pc.hip: S2K_OK = launched, 1 = not one of its shapes (the caller takes the generic kernels), < 0 = error
int launch_conv_pc(ConvP& p, hipStream_t st);
// conv_dma.hip: prologue-free 1x1 contractions on the LDS-DMA ring kernel; same return convention
int launch_conv_dma(ConvP& p, hipStream_t st);
// conv_q4.hip: prologue-free 1x1 contractions with the quad weight copy (p.wtq set); same return convention
int launch_conv_q4(ConvP& p, hipStream_t st);
// conv_bf16.hip (bf16-mixed plans only: p.wtb set): same return convention
int launch_conv_bf16(ConvP& p, hipStream_t st);

}  // namespace s2k
