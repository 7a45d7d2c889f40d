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
    int exp;                     // tuning builds only (S2K_WG_EXP): 1 = stage the first tile only, 2 = no combine, 4 = no MFMA loop
};

// wgrad_pc.hip: returns S2K_OK when it launched the stage, 1 when the shape is not one of its instantiations (the caller
// then takes the generic kernel), or a negative error code
int launch_wgrad_pc(WgradP& p, int mode, hipStream_t st);
// wgrad_bf16.hip (stages carrying S2K_FLAG_BF16 in bf16-mixed plans): same return convention
int launch_wgrad_bf16(WgradP& p, int mode, hipStream_t st);

// bijective remap: consecutive logical tiles land on the same XCD (hardware deals blocks round-robin over the 8 XCDs;
// which XCD is irrelevant, only that ids congruent mod 8 share one)
__device__ __forceinline__ int wg_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace s2k
