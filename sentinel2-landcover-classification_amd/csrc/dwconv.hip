// Depthwise KxK convolution (k in {3,5}, stride in {1,2}, TF-"SAME" asymmetric padding) — the
// HBM-bound part of every MBConv block (reference efficientnet_unet.py:335-352 via
// Conv2dSamePadding.forward :288-297, groups = channels) — forward, data gradient (+ fused act' and
// BatchNorm-backward sums of the producer) and weight gradient.
//
// One workgroup owns PPB consecutive (b,c) planes x a band of rows.  The band (+ halo) is read ONCE from
// HBM through bounds-checked buffer loads (zero padding = out-of-range offset), normalised + SiLU'd on
// the fly (the producing conv stored raw values) and staged in LDS.  Compute is register-blocked: a lane
// produces 4 consecutive outputs of a row from K x (K + 3*S) LDS values; small planes are packed several
// per wave (LPP lanes per plane), so per-plane sums (BN statistics of the output, BN-backward sums of
// the input gradient, the K*K weight-gradient sums) are log2(LPP) shuffle steps + one atomic per plane.
// K and S are compile-time.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace s2k {

struct DwP {
    const float* x;      // fwd/wgrad: input;  dgrad: XRAW (raw producer output) or null
    const float* bnv;    // [4][C] of the input's BatchNorm (scale, shift, mean, invstd) or null
    const float* w;      // [C][K][K]
    const float* dy;     // dgrad/wgrad
    float* out;          // fwd: Y; dgrad: G; wgrad: DW (atomics)
    double* stats;       // fwd: [R][2][C] sum/sumsq of Y;  dgrad: [R][2][C] sum g, sum g*xhat
    int B, C, H, W, K, S, PT, PL, HO, WO, pro, beta, nrep;
    int PPB, RT, bands, IRt, LW;     // tiling: planes per workgroup, rows per band, LDS rows / row stride
    int XG, LPP, lwp_shift;          // 4-wide x groups per row, lanes per plane, log2(pow2ceil(LW)) capped at 6
    int vps_shift;                   // log2(pow2ceil(LW / 4)) capped at 6 (vector stager)
    float inv_xg;                    // 1 / XG
    int bloop, cgroups;              // wgrad on small planes: images per workgroup (0 = off), channel groups
    BnFold fold;                     // fwd: BN_FINALIZE of the input's BatchNorm folded in (fold.stats != nullptr)
};

// Index arithmetic of the kernels below.  A 64-bit `plane % C` is ~150 instructions and a 32-bit division by a run-time value
// ~25; on the 8x8 / 16x16 layers (a few hundred instructions per wave in all) they were a third of the kernel.  Planes fit
// 32 bits (the launchers refuse B * C >= 2^31), and quotients of small non-negative ints come from one multiplication by a
// float reciprocal (exact for numerators < 2^22: the error of the product is far below the 0.5 / divisor margin).
__device__ __forceinline__ int chan_of(int64_t plane, int C) { return (int)((uint32_t)plane % (uint32_t)C); }
__device__ __forceinline__ int fdiv(int num, float inv) { return (int)(((float)num + 0.5f) * inv); }

// Scalar stager (any width / column origin): rows [row0, row0 + nrows) x LW columns of PPB planes of `src` into
// tile[pl][rr][LW]; element (rr, cc) is src row (row0 + rr), column (col0 + cc), zero outside the image.
template <int PRO>
__device__ __forceinline__ void stage_band(const DwP& p, const float* src, int Hs, int Ws, float* tile, int64_t pl0,
                                           int64_t nplanes, int nrows, int row0, int col0, const float* bnl = nullptr) {
    const rsrc_t rs = make_rsrc(src, nplanes * Hs * Ws * 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lwp = 1 << p.lwp_shift;                 // pow2 >= LW (capped at 64)
    const int rpw = 64 >> p.lwp_shift;                // rows per wave pass
    const int sub = lane >> p.lwp_shift, cc0 = lane & (lwp - 1);
    const int total_rows = p.PPB * nrows;
    const float inv_nrows = 1.0f / (float)nrows;
    if (PRO == S2K_PRO_NONE && p.LW <= lwp) {
        // a tile row fits one pass of a lane group: the loads of U row groups are issued before the first one is used
        // (see stage_band_v4)
        constexpr int U = 4;
        const int ix = col0 + cc0;
        const bool col_ok = cc0 < p.LW && ix >= 0 && ix < Ws;
        for (int rbase = wave * rpw; rbase < total_rows; rbase += 4 * rpw * U) {
            float v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = rbase + u * 4 * rpw + sub;
                const int pl = fdiv(row, inv_nrows), rr = row - pl * nrows;
                const int64_t plane = pl0 + pl;
                const int iy = row0 + rr;
                const bool ok = row < total_rows && plane < nplanes && iy >= 0 && iy < Hs && col_ok;
                v[u] = bload(rs, ok ? (uint32_t)((plane * Hs + iy) * Ws + ix) * 4u : BUF_OOB);
                dst[u] = (row < total_rows && cc0 < p.LW) ? (pl * nrows + rr) * p.LW + cc0 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) tile[dst[u]] = v[u];
        }
        return;
    }
    for (int rbase = wave * rpw; rbase < total_rows; rbase += 4 * rpw) {
        const int row = rbase + sub;
        const int pl = fdiv(row, inv_nrows), rr = row - pl * nrows;
        const int64_t plane = pl0 + pl;
        const bool rok = row < total_rows && plane < nplanes;
        const int iy = row0 + rr;
        const bool yok = rok && iy >= 0 && iy < Hs;
        float sc = 1.0f, sh = 0.0f;
        if (PRO != S2K_PRO_NONE) {
            if (bnl) {           // folded BN_FINALIZE: this workgroup's own {scale, shift}[PPB] in LDS
                sc = bnl[rok ? pl : 0];
                sh = bnl[p.PPB + (rok ? pl : 0)];
            } else {
                const int c = chan_of(rok ? plane : 0, p.C);
                sc = p.bnv[c];
                sh = p.bnv[p.C + c];
            }
        }
        const uint32_t rowoff = (uint32_t)((plane * Hs + iy) * Ws) * 4u;
        for (int cc = cc0; cc < p.LW; cc += lwp) {
            const int ix = col0 + cc;
            const bool ok = yok && ix >= 0 && ix < Ws;
            float v = bload(rs, ok ? rowoff + (uint32_t)ix * 4u : BUF_OOB);
            if (PRO != S2K_PRO_NONE) v = ok ? apply_pro_c<PRO>(v, sc, sh) : 0.0f;
            if (row < total_rows) tile[(pl * nrows + rr) * p.LW + cc] = v;
        }
    }
}

// Vector stager of the register-blocked kernels: LW % 4 == 0 and tile column cc holds source column cc - 4, so that
// source column 0 sits on a 16-byte boundary: a lane moves one aligned float4 HBM -> LDS (ds_write_b128); the 4-column
// halo on the left and the columns past the row end are zeros (or real data where the source row is wider).
template <int PRO>
__device__ __forceinline__ void stage_band_v4(const DwP& p, const float* src, int Hs, int Ws, float* tile, int64_t pl0,
                                              int64_t nplanes, int nrows, int row0, const float* bnl = nullptr) {
    const rsrc_t rs = make_rsrc(src, nplanes * Hs * Ws * 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int VPR = p.LW >> 2;                          // float4 slots per tile row
    const int vshift = p.vps_shift;                     // log2(pow2 >= VPR), capped at 6
    const int vps = 1 << vshift, rpw = 64 >> vshift;
    const int sub = lane >> vshift, v0 = lane & (vps - 1);
    const int total_rows = p.PPB * nrows;
    const float inv_nrows = 1.0f / (float)nrows;
    const bool vec = (Ws & 3) == 0;
    if (vec && VPR <= vps) {
        // A tile row fits one pass of a lane group: issue the loads of U row groups before the first one is used.  (The plain
        // loop below waits for each load in turn - six dependent HBM round trips per workgroup on the 8x8 planes, 20 us for a
        // tensor that a copy streams in 4.)
        constexpr int U = 4;
        const int ix = 4 * v0 - 4;
        const bool col_ok = v0 < VPR && ix >= 0 && ix < Ws;
        for (int rbase = wave * rpw; rbase < total_rows; rbase += 4 * rpw * U) {
            f32x4 v[U];
            float sc[U], sh[U];
            int dst[U];
            bool okv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = rbase + u * 4 * rpw + sub;
                const int pl = fdiv(row, inv_nrows), rr = row - pl * nrows;
                const int64_t plane = pl0 + pl;
                const bool rok = row < total_rows && plane < nplanes;
                const int iy = row0 + rr;
                okv[u] = rok && iy >= 0 && iy < Hs && col_ok;
                sc[u] = 1.0f;
                sh[u] = 0.0f;
                if (PRO != S2K_PRO_NONE) {
                    if (bnl) {
                        sc[u] = bnl[rok ? pl : 0];
                        sh[u] = bnl[p.PPB + (rok ? pl : 0)];
                    } else {
                        const int c = chan_of(rok ? plane : 0, p.C);
                        sc[u] = p.bnv[c];
                        sh[u] = p.bnv[p.C + c];
                    }
                }
                const uint32_t rowoff = (uint32_t)((plane * Hs + iy) * Ws) * 4u;
                v[u] = bload4(rs, okv[u] ? rowoff + (uint32_t)ix * 4u : BUF_OOB);
                dst[u] = (row < total_rows && v0 < VPR) ? (pl * nrows + rr) * p.LW + 4 * v0 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = okv[u] ? apply_pro_c<PRO>(v[u][j], sc[u], sh[u]) : 0.0f;
                }
                if (dst[u] >= 0) *reinterpret_cast<f32x4*>(tile + dst[u]) = v[u];
            }
        }
        return;
    }
    for (int rbase = wave * rpw; rbase < total_rows; rbase += 4 * rpw) {
        const int row = rbase + sub;
        const int pl = fdiv(row, inv_nrows), rr = row - pl * nrows;
        const int64_t plane = pl0 + pl;
        const bool rok = row < total_rows && plane < nplanes;
        const int iy = row0 + rr;
        const bool yok = rok && iy >= 0 && iy < Hs;
        float sc = 1.0f, sh = 0.0f;
        if (PRO != S2K_PRO_NONE) {
            if (bnl) {           // folded BN_FINALIZE: this workgroup's own {scale, shift}[PPB] in LDS
                sc = bnl[rok ? pl : 0];
                sh = bnl[p.PPB + (rok ? pl : 0)];
            } else {
                const int c = chan_of(rok ? plane : 0, p.C);
                sc = p.bnv[c];
                sh = p.bnv[p.C + c];
            }
        }
        const uint32_t rowoff = (uint32_t)((plane * Hs + iy) * Ws) * 4u;
        for (int vi = v0; vi < VPR; vi += vps) {
            const int ix = 4 * vi - 4;
            f32x4 v;
            if (vec) {
                const bool ok = yok && ix >= 0 && ix < Ws;
                v = bload4(rs, ok ? rowoff + (uint32_t)ix * 4u : BUF_OOB);
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ok ? apply_pro_c<PRO>(v[j], sc, sh) : 0.0f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = yok && ix + j >= 0 && ix + j < Ws;
                    float e = bload(rs, ok ? rowoff + (uint32_t)(ix + j) * 4u : BUF_OOB);
                    if (PRO != S2K_PRO_NONE) e = ok ? apply_pro_c<PRO>(e, sc, sh) : 0.0f;
                    v[j] = e;
                }
            }
            if (row < total_rows) *reinterpret_cast<f32x4*>(tile + (pl * nrows + rr) * p.LW + 4 * vi) = v;
        }
    }
}

// one row window of an item: NV aligned float4 LDS reads (ds_read_b128) -> v[4 * NV]
template <int NV>
__device__ __forceinline__ void read_window(const float* t, float* v) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(t + 4 * i);
        v[4 * i + 0] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
    }
}

// ---- forward -------------------------------------------------------------------------------------
// PL = left padding (compile time: the column window of an item then starts at the constant offset 4 - PL of an aligned
// float4 group, so every LDS operand index below is a constant)
template <int K, int S, int PL, int PRO>
__global__ void __launch_bounds__(NTHREADS) dwconv_fwd_kernel(const DwP p) {
    constexpr int NX = 3 * S + K;         // values of one row needed for 4 outputs
    constexpr int O0 = 4 - PL;            // first of them inside the aligned window
    constexpr int NV = (O0 + NX + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [PPB][IRt][LW]
    float* wsm = smem + p.PPB * p.IRt * p.LW;         // [PPB][K*K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int yo0 = band * p.RT;
    const int rows = min(p.RT, p.HO - yo0);
    float* bnl = nullptr;
    if (PRO != S2K_PRO_NONE && p.fold.stats) {
        // BN_FINALIZE of the input's BatchNorm, for this workgroup's planes; the workgroup that holds sample 0 / band 0 of a
        // channel publishes BNV and updates the running statistics
        bnl = wsm + p.PPB * K * K;                    // [2][PPB]
        if (p.PPB <= 4) {
            if (wave < p.PPB) {
                const int64_t plane = pl0 + wave;
                float sc = 1.0f, sh = 0.0f;
                if (plane < nplanes) bn_fold_wave(p.fold, p.C, chan_of(plane, p.C), plane < p.C && band == 0, sc, sh);
                if (lane == 0) { bnl[wave] = sc; bnl[p.PPB + wave] = sh; }
            }
        } else {
            for (int pl = tid; pl < p.PPB; pl += NTHREADS) {
                const int64_t plane = pl0 + pl;
                float sc = 1.0f, sh = 0.0f;
                if (plane < nplanes) bn_fold_thread(p.fold, p.C, chan_of(plane, p.C), plane < p.C && band == 0, sc, sh);
                bnl[pl] = sc;
                bnl[p.PPB + pl] = sh;
            }
        }
        __syncthreads();
    }
    stage_band_v4<PRO>(p, p.x, p.H, p.W, tile, pl0, nplanes, p.IRt, yo0 * S - p.PT, bnl);
    for (int idx = tid; idx < p.PPB * K * K; idx += NTHREADS) {
        const int pl = idx / (K * K);
        const int64_t plane = pl0 + pl;
        wsm[idx] = plane < nplanes ? p.w[chan_of(plane, p.C) * (K * K) + (idx - pl * (K * K))] : 0.0f;
    }
    __syncthreads();

    const int items = rows * p.XG;                    // 4-wide output groups per plane
    const int lpp = p.LPP, ppw = 64 / lpp;            // lanes per plane, planes per wave
    const int li = lane & (lpp - 1), lp = lane / lpp;
    double* st = p.stats ? p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.C : nullptr;
    const bool vec4 = (p.WO & 3) == 0;
    // big planes (fewer than 4 per workgroup): wpp waves share one plane and interleave its items
    const int wpp = (lpp == 64 && p.PPB < 4) ? 4 / p.PPB : 1;
    const int it0 = li + lpp * (wave % wpp), itstep = lpp * wpp;
    for (int pg = (wave / wpp) * ppw; pg < p.PPB; pg += (4 / wpp) * ppw) {
        const int pl = pg + lp;
        const int64_t plane = pl0 + pl;
        const bool pok = pl < p.PPB && plane < nplanes;
        float s = 0.0f, q = 0.0f;
        if (pok) {
            float wk[K * K];
#pragma unroll
            for (int i = 0; i < K * K; ++i) wk[i] = wsm[pl * K * K + i];
            const float* tp = tile + pl * p.IRt * p.LW;
            for (int it = it0; it < items; it += itstep) {
                const int r = fdiv(it, p.inv_xg), xg = it - r * p.XG;
                const float* t0 = tp + (r * S) * p.LW + xg * 4 * S;
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    float v[4 * NV];
                    read_window<NV>(t0 + ky * p.LW, v);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = fmaf(wk[ky * K + kx], v[O0 + j * S + kx], o[j]);
                }
                const int xo = xg * 4;
                float* dst = p.out + plane * p.HO * p.WO + (int64_t)(yo0 + r) * p.WO + xo;
                if (vec4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s += o[j]; q = fmaf(o[j], o[j], q); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (xo + j < p.WO) { dst[j] = o[j]; s += o[j]; q = fmaf(o[j], o[j], q); }
                }
            }
        }
        if (st) {
            s = group_sum(s, lpp);
            q = group_sum(q, lpp);
            if (li == 0 && pok) {
                const int c = chan_of(plane, p.C);
                atomic_add_d(st + c, (double)s);
                atomic_add_d(st + p.C + c, (double)q);
            }
        }
    }
}

// ---- forward on small square planes (8 x 8, 16 x 16, 32 x 32; stride 1: the deep MBConv blocks) --------------------------------------------
// The band kernel above stages a group of planes, synchronises the workgroup, computes, stores - one pass per workgroup, so its HBM
// round trip, the SiLU prologue, the K*K multiply-adds and the stores follow one another and only other workgroups on the CU overlap
// them (16 x 16, k = 5: 42 us for 69 MB = 1.65 TB/s; the bare structure without prologue and statistics: 26 us; a copy: 11 us).
// Here ONE WAVE owns a channel and walks (a chunk of) the batch with no workgroup barrier at all: the plane after the current one is
// already in flight (one 16-byte load per lane = the whole 16 x 16 plane per instruction; four 8 x 8 planes), the current one goes
// through the prologue into a wave-private LDS tile whose zero halo is written once, every lane computes 4 outputs from K rows of
// three aligned 16-byte LDS reads, stores them and keeps the channel's BatchNorm sums in registers: one f64 atomic pair per wave,
// the input BatchNorm's fold (opdefs.FOLD_*) and the K*K weights once per wave.  LDS operations of one wave execute in order, so
// the tile needs no barrier; two tiles alternate so that a pass never overwrites what the previous one may still be reading.
// R < W (64 x 64 and 128 x 128 planes): the wave's unit of work is a band of R rows of one plane instead of a whole plane; the K - 1
// halo rows above / below the band come with one extra (bounds-checked) 16-byte load per lane and are rewritten for every band.
template <int K, int PRO, int W, int R = W>
__global__ void __launch_bounds__(NTHREADS) dwconv_fwd_plane_kernel(const DwP p, int bchunk) {
    constexpr int PADK = (K - 1) / 2;
    constexpr bool BAND = R != W;
    constexpr int NB = W / R;                      // bands per plane
    constexpr int GPP = R * W / 4;                 // 4-pixel groups per work item
    constexpr int NG = GPP > 64 ? GPP / 64 : 1;    // groups per lane
    constexpr int LPP = GPP > 64 ? 64 : GPP;       // lanes per item: 64, or 16 (8 x 8 planes)
    constexpr int PW = 64 / LPP;                   // items per wave pass
    constexpr int XGW = W / 4;
    constexpr int HG = BAND ? (K - 1) * XGW : 0;   // halo groups of a band (<= 64: one per lane)
    constexpr int TH = R + K - 1, TW = W + 8;      // tile rows; columns: image column x sits at 4 + x
    constexpr int TILE = TH * TW;
    constexpr int HW = W * W;
    static_assert(HG <= 64 && (!BAND || PW == 1) && W % R == 0, "band geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][2][PW][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= p.C) return;                          // (no workgroup barrier anywhere in this kernel)
    // work items of this channel: (image, band) pairs; this wave takes items [i_lo, i_hi)
    const int n_items = p.B * NB;
    const int i_lo = blockIdx.y * bchunk, i_hi = min(n_items, i_lo + bchunk);
    float sc = 1.0f, sh = 0.0f;
    if (PRO != S2K_PRO_NONE) {
        if (p.fold.stats) bn_fold_wave(p.fold, p.C, c, blockIdx.y == 0, sc, sh);
        else { sc = p.bnv[c]; sh = p.bnv[p.C + c]; }
    }
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = p.w[c * (K * K) + i];
    float* tiles = smem + wave * (2 * PW * TILE);
    for (int i = lane; i < 2 * PW * TILE; i += 64) tiles[i] = 0.0f;
    const int sub = lane / LPP, li = lane % LPP;
    const int64_t bstride = (int64_t)p.C * HW, coff = (int64_t)c * HW;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // halo group of this lane: tile row / image row offset relative to the band's first row
    const int hrow = lane / XGW, hxg = lane % XGW;                            // (used when lane < HG)
    const int h_trow = hrow < PADK ? hrow : R + hrow;                         // tile row
    const int h_dy = hrow < PADK ? hrow - PADK : R + hrow - PADK;             // image row - y0
    auto item_off = [&](int it, int& y0) -> int64_t {                         // element offset of an item's first row
        const int b = BAND ? it / NB : it;
        y0 = BAND ? (it - b * NB) * R : 0;
        return (int64_t)b * bstride + coff + (int64_t)y0 * W;
    };
    f32x4 cur[NG], hcur = zero;
    {
        int y0;
        const int it = i_lo + sub;
        const int64_t off = item_off(it < i_hi ? it : i_lo, y0);
#pragma unroll
        for (int g = 0; g < NG; ++g) cur[g] = (it < i_hi) ? *reinterpret_cast<const f32x4*>(p.x + off + 4 * li + 256 * g) : zero;
        if (BAND && lane < HG && it < i_hi && y0 + h_dy >= 0 && y0 + h_dy < W) hcur = *reinterpret_cast<const f32x4*>(p.x + off + (int64_t)h_dy * W + 4 * hxg);
    }
    float s = 0.0f, q = 0.0f;
    int buf = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PW, buf ^= 1) {
        const int it = i0 + sub;
        const bool ok = it < i_hi;
        int y0;
        const int64_t off = item_off(ok ? it : i_lo, y0);
        f32x4 nxt[NG], hnxt = zero;
        {
            int y1;
            const bool more = it + PW < i_hi;
            const int64_t offn = item_off(more ? it + PW : i_lo, y1);
#pragma unroll
            for (int g = 0; g < NG; ++g) nxt[g] = more ? *reinterpret_cast<const f32x4*>(p.x + offn + 4 * li + 256 * g) : zero;   // in flight during this pass
            if (BAND && lane < HG && more && y1 + h_dy >= 0 && y1 + h_dy < W) hnxt = *reinterpret_cast<const f32x4*>(p.x + offn + (int64_t)h_dy * W + 4 * hxg);
        }
        float* t = tiles + (buf * PW + sub) * TILE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            f32x4 v = cur[g];
            if (PRO != S2K_PRO_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
            }
            if (!ok) v = zero;
            *reinterpret_cast<f32x4*>(t + (r + PADK) * TW + 4 + 4 * xg) = v;
        }
        if (BAND && lane < HG) {
            f32x4 v = hcur;
            const bool inside = ok && y0 + h_dy >= 0 && y0 + h_dy < W;        // the reference pads ACTIVATED maps with zeros
            if (PRO != S2K_PRO_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
            }
            if (!inside) v = zero;
            *reinterpret_cast<f32x4*>(t + h_trow * TW + 4 + 4 * hxg) = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            const float* t0 = t + r * TW + 4 * xg;       // rows r .. r + K - 1, tile columns 4 xg .. 4 xg + 11 = image columns 4 xg - 4 ..
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float win[12];
                read_window<3>(t0 + ky * TW, win);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(wk[ky * K + kx], win[4 - PADK + j + kx], o[j]);
            }
            if (ok) {
                *reinterpret_cast<f32x4*>(p.out + off + 4 * li + 256 * g) = f32x4{o[0], o[1], o[2], o[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { s += o[j]; q = fmaf(o[j], o[j], q); }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
        hcur = hnxt;
    }
    if (p.stats) {
        const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
        if (lane == 0) {
            double* st = p.stats + (int64_t)((blockIdx.x + blockIdx.y) % p.nrep) * 2 * p.C;
            atomic_add_d(st + c, sd);
            atomic_add_d(st + p.C + c, qd);
        }
    }
}

// ---- weight gradient on square planes at stride 1: the wave-per-channel scheme of dwconv_fwd_plane_kernel ----------------------------
// The activated input goes through the wave-private tile exactly as in the forward; the lane's dY pixels stay in registers, and every
// lane keeps the K * K tap sums of its pixels over ALL the items the wave walks: one wave reduction and one float atomic per tap at
// the very end (the band kernel: per plane group).
template <int K, int PRO, int W, int R = W>
__global__ void __launch_bounds__(NTHREADS) dwconv_wgrad_plane_kernel(const DwP p, int bchunk) {
    constexpr int PADK = (K - 1) / 2;
    constexpr bool BAND = R != W;
    constexpr int NB = W / R, GPP = R * W / 4, NG = GPP > 64 ? GPP / 64 : 1, LPP = GPP > 64 ? 64 : GPP, PW = 64 / LPP, XGW = W / 4;
    constexpr int HG = BAND ? (K - 1) * XGW : 0;
    constexpr int TH = R + K - 1, TW = W + 8, TILE = TH * TW, HW = W * W;
    static_assert(HG <= 64 && (!BAND || PW == 1) && W % R == 0, "band geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][2][PW][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= p.C) return;
    const int n_items = p.B * NB;
    const int i_lo = blockIdx.y * bchunk, i_hi = min(n_items, i_lo + bchunk);
    float sc = 1.0f, sh = 0.0f;
    if (PRO != S2K_PRO_NONE) { sc = p.bnv[c]; sh = p.bnv[p.C + c]; }
    float* tiles = smem + wave * (2 * PW * TILE);
    for (int i = lane; i < 2 * PW * TILE; i += 64) tiles[i] = 0.0f;
    const int sub = lane / LPP, li = lane % LPP;
    const int64_t bstride = (int64_t)p.C * HW, coff = (int64_t)c * HW;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int hrow = lane / XGW, hxg = lane % XGW;
    const int h_trow = hrow < PADK ? hrow : R + hrow;
    const int h_dy = hrow < PADK ? hrow - PADK : R + hrow - PADK;
    auto item_off = [&](int it, int& y0) -> int64_t {
        const int b = BAND ? it / NB : it;
        y0 = BAND ? (it - b * NB) * R : 0;
        return (int64_t)b * bstride + coff + (int64_t)y0 * W;
    };
    f32x4 cur[NG], dcur[NG], hcur = zero;
    {
        int y0;
        const int it = i_lo + sub;
        const bool ok0 = it < i_hi;
        const int64_t off = item_off(ok0 ? it : i_lo, y0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cur[g] = ok0 ? *reinterpret_cast<const f32x4*>(p.x + off + 4 * li + 256 * g) : zero;
            dcur[g] = ok0 ? *reinterpret_cast<const f32x4*>(p.dy + off + 4 * li + 256 * g) : zero;
        }
        if (BAND && lane < HG && ok0 && y0 + h_dy >= 0 && y0 + h_dy < W) hcur = *reinterpret_cast<const f32x4*>(p.x + off + (int64_t)h_dy * W + 4 * hxg);
    }
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.0f;
    int buf = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PW, buf ^= 1) {
        const int it = i0 + sub;
        const bool ok = it < i_hi, more = it + PW < i_hi;
        int y0, y1;
        (void)item_off(ok ? it : i_lo, y0);
        const int64_t offn = item_off(more ? it + PW : i_lo, y1);
        f32x4 nxt[NG], dnxt[NG], hnxt = zero;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            nxt[g] = more ? *reinterpret_cast<const f32x4*>(p.x + offn + 4 * li + 256 * g) : zero;
            dnxt[g] = more ? *reinterpret_cast<const f32x4*>(p.dy + offn + 4 * li + 256 * g) : zero;
        }
        if (BAND && lane < HG && more && y1 + h_dy >= 0 && y1 + h_dy < W) hnxt = *reinterpret_cast<const f32x4*>(p.x + offn + (int64_t)h_dy * W + 4 * hxg);
        float* t = tiles + (buf * PW + sub) * TILE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            f32x4 v = cur[g];
            if (PRO != S2K_PRO_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
            }
            if (!ok) v = zero;
            *reinterpret_cast<f32x4*>(t + (r + PADK) * TW + 4 + 4 * xg) = v;
        }
        if (BAND && lane < HG) {
            f32x4 v = hcur;
            if (PRO != S2K_PRO_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
            }
            if (!(ok && y0 + h_dy >= 0 && y0 + h_dy < W)) v = zero;
            *reinterpret_cast<f32x4*>(t + h_trow * TW + 4 + 4 * hxg) = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            const float* t0 = t + r * TW + 4 * xg;
            const f32x4 d = dcur[g];                                  // zero when the item does not exist
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float win[12];
                read_window<3>(t0 + ky * TW, win);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * K + kx] = fmaf(d[j], win[4 - PADK + j + kx], acc[ky * K + kx]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) { cur[g] = nxt[g]; dcur[g] = dnxt[g]; }
        hcur = hnxt;
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) atomicAdd(p.out + c * (K * K) + i, v);
    }
}

// ---- forward at stride 2 (even sizes, TF-SAME: K - 2 pad rows / columns, the larger half below / right): the same wave-per-channel
// scheme; a work item = RO output rows = 2 RO input rows (+ K - 2 halo rows), 4 input groups and 1 output group per lane
template <int K, int PRO, int WO, int RO>
__global__ void __launch_bounds__(NTHREADS) dwconv_fwd_plane_s2_kernel(const DwP p, int bchunk) {
    constexpr int WI = 2 * WO, PT = (K - 2) / 2, PB = K - 2 - PT;     // pads: top / left PT, bottom / right PB
    constexpr int NB = WO / RO;                    // bands per plane (RO == WO: whole planes)
    constexpr bool BAND = NB > 1;
    constexpr int GO = RO * WO / 4;                // output groups per item: 64, or 16 (8 x 8 outputs)
    constexpr int LPP = GO, PW = 64 / LPP;
    constexpr int XGO = WO / 4, XGI = WI / 4;
    constexpr int NGI = 2 * RO * XGI / LPP;        // input groups per lane (= 4)
    constexpr int HGT = BAND ? (K - 2) * XGI : 0;  // halo groups of a band
    constexpr int NH = (HGT + 63) / 64;
    constexpr int TH = 2 * RO + K - 2, TW = WI + 8, TILE = TH * TW;
    constexpr int HWI = WI * WI, HWO = WO * WO;
    static_assert(NGI == 4 && (!BAND || PW == 1), "stride-2 plane geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][2][PW][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= p.C) return;
    const int n_items = p.B * NB;
    const int i_lo = blockIdx.y * bchunk, i_hi = min(n_items, i_lo + bchunk);
    float sc = 1.0f, sh = 0.0f;
    if (PRO != S2K_PRO_NONE) {
        if (p.fold.stats) bn_fold_wave(p.fold, p.C, c, blockIdx.y == 0, sc, sh);
        else { sc = p.bnv[c]; sh = p.bnv[p.C + c]; }
    }
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = p.w[c * (K * K) + i];
    float* tiles = smem + wave * (2 * PW * TILE);
    for (int i = lane; i < 2 * PW * TILE; i += 64) tiles[i] = 0.0f;
    const int sub = lane / LPP, li = lane % LPP;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int64_t istride = (int64_t)p.C * HWI, ostride = (int64_t)p.C * HWO;
    // item -> (image, first output row)
    auto split = [&](int it, int& b, int& yo0) { b = BAND ? it / NB : it; yo0 = BAND ? (it - b * NB) * RO : 0; };
    // halo group h (0 .. HGT-1) of a band: tile row / input row relative to the band's first input row 2 yo0
    auto halo_geo = [&](int h, int& trow, int& dy, int& xg) {
        const int hr = h / XGI; xg = h % XGI;
        trow = hr < PT ? hr : 2 * RO + hr;
        dy = hr < PT ? hr - PT : 2 * RO + hr - PT;
    };
    f32x4 cur[NGI], hcur[NH > 0 ? NH : 1];
    auto fetch = [&](int it, bool valid, f32x4 (&v)[NGI], f32x4 (&hv)[NH > 0 ? NH : 1]) {
        int b, yo0;
        split(valid ? it : i_lo, b, yo0);
        const float* src = p.x + (int64_t)b * istride + (int64_t)c * HWI + (int64_t)(2 * yo0) * WI;
#pragma unroll
        for (int g = 0; g < NGI; ++g) v[g] = valid ? *reinterpret_cast<const f32x4*>(src + 4 * (li + LPP * g)) : zero;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            hv[k] = zero;
            const int h = lane + 64 * k;
            if (BAND && h < HGT && valid) {
                int trow, dy, xg;
                halo_geo(h, trow, dy, xg);
                const int iy = 2 * yo0 + dy;
                if (iy >= 0 && iy < WI) hv[k] = *reinterpret_cast<const f32x4*>(src + (int64_t)dy * WI + 4 * xg);
            }
        }
    };
    fetch(i_lo + sub, i_lo + sub < i_hi, cur, hcur);
    float s = 0.0f, q = 0.0f;
    int buf = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PW, buf ^= 1) {
        const int it = i0 + sub;
        const bool ok = it < i_hi;
        f32x4 nxt[NGI], hnxt[NH > 0 ? NH : 1];
        fetch(it + PW, it + PW < i_hi, nxt, hnxt);                       // in flight during this pass
        int b, yo0;
        split(ok ? it : i_lo, b, yo0);
        float* t = tiles + (buf * PW + sub) * TILE;
#pragma unroll
        for (int g = 0; g < NGI; ++g) {
            const int gi = li + LPP * g, r = gi / XGI, xg = gi % XGI;
            f32x4 v = cur[g];
            if (PRO != S2K_PRO_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
            }
            if (!ok) v = zero;
            *reinterpret_cast<f32x4*>(t + (r + PT) * TW + 4 + 4 * xg) = v;
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int h = lane + 64 * k;
            if (BAND && h < HGT) {
                int trow, dy, xg;
                halo_geo(h, trow, dy, xg);
                const int iy = 2 * yo0 + dy;
                f32x4 v = hcur[k];
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_pro_c<PRO>(v[j], sc, sh);
                }
                if (!(ok && iy >= 0 && iy < WI)) v = zero;               // the reference pads ACTIVATED maps with zeros
                *reinterpret_cast<f32x4*>(t + trow * TW + 4 + 4 * xg) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int r = li / XGO, xg = li % XGO;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            const float* t0 = t + (2 * r) * TW + 8 * xg;                 // input rows 2r .., tile columns 8 xg .. 8 xg + 15
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float win[16];
                read_window<4>(t0 + ky * TW, win);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(wk[ky * K + kx], win[4 - PT + 2 * j + kx], o[j]);
            }
            if (ok) {
                *reinterpret_cast<f32x4*>(p.out + (int64_t)b * ostride + (int64_t)c * HWO + (int64_t)(yo0 + r) * WO + 4 * xg) = f32x4{o[0], o[1], o[2], o[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { s += o[j]; q = fmaf(o[j], o[j], q); }
            }
        }
#pragma unroll
        for (int g = 0; g < NGI; ++g) cur[g] = nxt[g];
#pragma unroll
        for (int k = 0; k < NH; ++k) hcur[k] = hnxt[k];
    }
    if (p.stats) {
        const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
        if (lane == 0) {
            double* st = p.stats + (int64_t)((blockIdx.x + blockIdx.y) % p.nrep) * 2 * p.C;
            atomic_add_d(st + c, sd);
            atomic_add_d(st + p.C + c, qd);
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------
template <int K, int S, int PL, int PRO>
__global__ void __launch_bounds__(NTHREADS) dwconv_wgrad_kernel(const DwP p) {
    constexpr int NX = 3 * S + K;
    constexpr int O0 = 4 - PL;
    constexpr int NV = (O0 + NX + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;  // [PPB][IRt][LW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int yo0 = band * p.RT;
    const int rows = min(p.RT, p.HO - yo0);
    const rsrc_t rdy = make_rsrc(p.dy, nplanes * p.HO * p.WO * 4);
    const int items = rows * p.XG;
    const int lpp = p.LPP, ppw = 64 / lpp;
    const int li = lane & (lpp - 1), lp = lane / lpp;
    const bool vec4 = (p.WO & 3) == 0;
    if (p.bloop > 0) {
        // Small planes (8x8 ... 32x32 maps with hundreds of channels): a workgroup owns PPB CHANNELS and walks over a range
        // of images with the tap sums kept in registers, then adds each (channel, tap) once.  One atomic per (image, channel,
        // tap) — what the plane-per-workgroup form below does — is 1.5 M nearly empty atomic instructions on a
        // 1824-channel 8x8 layer, and atomics execute at the memory side at one wave-instruction per ~50 ns per CU whatever
        // their lane count: that, not the arithmetic, was 90 % of such a layer's time.
        const int cg = blockIdx.x % p.cgroups, bs = blockIdx.x / p.cgroups;
        const int c0 = cg * p.PPB;
        const int b0 = bs * p.bloop, b1 = min(p.B, b0 + p.bloop);
        const int pl = wave * ppw + lp;                  // PPB == 4 * ppw: exactly one plane group per wave
        const bool cok = pl < p.PPB && c0 + pl < p.C;
        float acc[K * K];
#pragma unroll
        for (int i = 0; i < K * K; ++i) acc[i] = 0.0f;
        for (int b = b0; b < b1; ++b) {
            const int64_t plb = (int64_t)b * p.C + c0;
            stage_band_v4<PRO>(p, p.x, p.H, p.W, tile, plb, (int64_t)(b + 1) * p.C, p.IRt, -p.PT);
            __syncthreads();
            if (cok) {
                const int64_t plane = plb + pl;
                const float* tp = tile + pl * p.IRt * p.LW;
                for (int it = li; it < items; it += lpp) {
                    const int r = fdiv(it, p.inv_xg), xg = it - r * p.XG;
                    const int xo = xg * 4;
                    const uint32_t goff = (uint32_t)(plane * p.HO * p.WO + (int64_t)r * p.WO + xo) * 4u;
                    float g[4];
                    if (vec4) {
                        const f32x4 gv = bload4(rdy, goff);
                        g[0] = gv[0]; g[1] = gv[1]; g[2] = gv[2]; g[3] = gv[3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j] = bload(rdy, xo + j < p.WO ? goff + 4u * j : BUF_OOB);
                    }
                    const float* t0 = tp + (r * S) * p.LW + xo * S;
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        float v[4 * NV];
                        read_window<NV>(t0 + ky * p.LW, v);
#pragma unroll
                        for (int kx = 0; kx < K; ++kx)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[ky * K + kx] = fmaf(g[j], v[O0 + j * S + kx], acc[ky * K + kx]);
                    }
                }
            }
            __syncthreads();
        }
        float* redc = smem + p.PPB * p.IRt * p.LW;       // [PPB][K*K]
#pragma unroll
        for (int i = 0; i < K * K; ++i) {
            const float v = group_sum(acc[i], lpp);
            if (li == 0 && pl < p.PPB) redc[pl * (K * K) + i] = cok ? v : 0.0f;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < p.PPB * K * K; idx += NTHREADS) {
            const int q = idx / (K * K);
            if (c0 + q < p.C) atomicAdd(p.out + (int64_t)(c0 + q) * (K * K) + (idx - q * (K * K)), redc[idx]);
        }
        return;
    }
    stage_band_v4<PRO>(p, p.x, p.H, p.W, tile, pl0, nplanes, p.IRt, yo0 * S - p.PT);
    __syncthreads();
    // big planes (fewer than 4 per workgroup): wpp waves share one plane and interleave its items
    const int wpp = (lpp == 64 && p.PPB < 4) ? 4 / p.PPB : 1;
    const int it0 = li + lpp * (wave % wpp), itstep = lpp * wpp;
    float* red = smem + p.PPB * p.IRt * p.LW;   // [4 waves][K*K] cross-wave combine (wpp > 1 only)
    for (int pgb = 0; pgb < p.PPB; pgb += (4 / wpp) * ppw) {
        const int pl = pgb + (wave / wpp) * ppw + lp;
        const int64_t plane = pl0 + pl;
        const bool pok = pl < p.PPB && plane < nplanes;
        float acc[K * K];
#pragma unroll
        for (int i = 0; i < K * K; ++i) acc[i] = 0.0f;
        if (pok) {
            const float* tp = tile + pl * p.IRt * p.LW;
            for (int it = it0; it < items; it += itstep) {
                const int r = fdiv(it, p.inv_xg), xg = it - r * p.XG;
                const int xo = xg * 4;
                const uint32_t goff = (uint32_t)(plane * p.HO * p.WO + (int64_t)(yo0 + r) * p.WO + xo) * 4u;
                float g[4];
                if (vec4) {
                    const f32x4 gv = bload4(rdy, goff);
                    g[0] = gv[0]; g[1] = gv[1]; g[2] = gv[2]; g[3] = gv[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = bload(rdy, xo + j < p.WO ? goff + 4u * j : BUF_OOB);
                }
                const float* t0 = tp + (r * S) * p.LW + xo * S;
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    float v[4 * NV];
                    read_window<NV>(t0 + ky * p.LW, v);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ky * K + kx] = fmaf(g[j], v[O0 + j * S + kx], acc[ky * K + kx]);
                }
            }
        }
        const int c = chan_of(pok ? plane : 0, p.C);
        if (wpp == 1) {
#pragma unroll
            for (int i = 0; i < K * K; ++i) {
                const float v = group_sum(acc[i], lpp);
                if (li == 0 && pok) atomicAdd(p.out + (int64_t)c * (K * K) + i, v);
            }
        } else {
            // one atomic per (plane, tap) per workgroup: same-address atomics serialise at the memory side
#pragma unroll
            for (int i = 0; i < K * K; ++i) {
                const float v = wave_sum(acc[i]);
                if (lane == 0) red[wave * (K * K) + i] = pok ? v : 0.0f;
            }
            __syncthreads();
            const int nslot = 4 / wpp;
            if ((int)threadIdx.x < nslot * K * K) {
                const int slot = threadIdx.x / (K * K), i = threadIdx.x - slot * (K * K);
                const int64_t pln = pl0 + pgb + slot;
                if (pgb + slot < p.PPB && pln < nplanes) {
                    float v = 0.0f;
                    for (int w = 0; w < wpp; ++w) v += red[(slot * wpp + w) * (K * K) + i];
                    atomicAdd(p.out + chan_of(pln, p.C) * (K * K) + i, v);
                }
            }
            __syncthreads();
        }
    }
}

// ---- data gradient, stride 1 (the common case): a correlation with the flipped kernel ------------------
// G[iy][ix] = sum_{ky,kx} w[ky][kx] * dY[iy + PT - ky][ix + PL - kx]; the LDS band holds dY rows
// [iy0 + PT - (K-1), ...); with a = K-1-ky, b = K-1-kx the window of 4 outputs starts at column 4*xg + (PL - (K-1)),
// i.e. at offset 4 - (K-1-PL) of the aligned group (PR = K-1-PL = the right padding of the forward).
template <int K, int PR, int PRO>
__global__ void __launch_bounds__(NTHREADS) dwconv_dgrad_s1_kernel(const DwP p) {
    constexpr int NX = 3 + K;
    constexpr int O0 = 4 - PR;
    constexpr int NV = (O0 + NX + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [PPB][IRt][LW]
    float* wsm = smem + p.PPB * p.IRt * p.LW;         // [PPB][K*K] flipped
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int iy0 = band * p.RT;
    const int rows = min(p.RT, p.H - iy0);
    stage_band_v4<S2K_PRO_NONE>(p, p.dy, p.HO, p.WO, tile, pl0, nplanes, p.IRt, iy0 + p.PT - (K - 1));
    for (int idx = tid; idx < p.PPB * K * K; idx += NTHREADS) {
        const int pl = idx / (K * K);
        const int64_t plane = pl0 + pl;
        wsm[idx] = plane < nplanes ? p.w[chan_of(plane, p.C) * (K * K) + (K * K - 1 - (idx - pl * (K * K)))] : 0.0f;
    }
    __syncthreads();
    const rsrc_t rxr = make_rsrc(p.x ? p.x : p.dy, p.x ? nplanes * p.H * p.W * 4 : 0);
    const rsrc_t rout = make_rsrc(p.out, nplanes * p.H * p.W * 4);
    const int items = rows * p.XG;
    const int lpp = p.LPP, ppw = 64 / lpp;
    const int li = lane & (lpp - 1), lp = lane / lpp;
    double* st = p.stats ? p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.C : nullptr;
    const bool vec4 = (p.W & 3) == 0;
    // big planes (fewer than 4 per workgroup): wpp waves share one plane and interleave its items
    const int wpp = (lpp == 64 && p.PPB < 4) ? 4 / p.PPB : 1;
    const int it0 = li + lpp * (wave % wpp), itstep = lpp * wpp;
    for (int pg = (wave / wpp) * ppw; pg < p.PPB; pg += (4 / wpp) * ppw) {
        const int pl = pg + lp;
        const int64_t plane = pl0 + pl;
        const bool pok = pl < p.PPB && plane < nplanes;
        const int c = chan_of(pok ? plane : 0, p.C);
        float s1 = 0.0f, s2 = 0.0f;
        if (pok) {
            float scale = 1.0f, shift = 0.0f, mean = 0.0f, invstd = 1.0f;
            if (PRO != S2K_PRO_NONE) { scale = p.bnv[c]; shift = p.bnv[p.C + c]; mean = p.bnv[2 * p.C + c]; invstd = p.bnv[3 * p.C + c]; }
            float wk[K * K];
#pragma unroll
            for (int i = 0; i < K * K; ++i) wk[i] = wsm[pl * K * K + i];
            const float* tp = tile + pl * p.IRt * p.LW;
            for (int it = it0; it < items; it += itstep) {
                const int r = fdiv(it, p.inv_xg), xg = it - r * p.XG;
                const int ix = xg * 4;
                const float* t0 = tp + r * p.LW + ix;
                const int64_t off = plane * p.H * p.W + (int64_t)(iy0 + r) * p.W + ix;
                const uint32_t boff = (uint32_t)off * 4u;
                float xr[4] = {0.f, 0.f, 0.f, 0.f}, ob[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec4) {       // issue the global loads before the LDS work
                    if (PRO != S2K_PRO_NONE) { const f32x4 t = bload4(rxr, boff); xr[0] = t[0]; xr[1] = t[1]; xr[2] = t[2]; xr[3] = t[3]; }
                    if (p.beta) { const f32x4 t = bload4(rout, boff); ob[0] = t[0]; ob[1] = t[1]; ob[2] = t[2]; ob[3] = t[3]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (PRO != S2K_PRO_NONE) xr[j] = bload(rxr, ix + j < p.W ? boff + 4u * j : BUF_OOB);
                        if (p.beta) ob[j] = bload(rout, ix + j < p.W ? boff + 4u * j : BUF_OOB);
                    }
                }
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < K; ++a) {
                    float v[4 * NV];
                    read_window<NV>(t0 + a * p.LW, v);
#pragma unroll
                    for (int b = 0; b < K; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = fmaf(wk[a * K + b], v[O0 + j + b], o[j]);
                }
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] *= act_grad(fmaf(xr[j], scale, shift), PRO);
                        if (ix + j < p.W) { s1 += o[j]; s2 = fmaf(o[j], (xr[j] - mean) * invstd, s2); }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += ob[j];
                float* dst = p.out + off;
                if (vec4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ix + j < p.W) dst[j] = o[j];
                }
            }
        }
        if (st) {
            s1 = group_sum(s1, lpp);
            s2 = group_sum(s2, lpp);
            if (li == 0 && pok) {
                atomic_add_d(st + c, (double)s1);
                atomic_add_d(st + p.C + c, (double)s2);
            }
        }
    }
}

// ---- data gradient on small square planes (8 x 8, 16 x 16, 32 x 32; stride 1): the wave-per-channel scheme of dwconv_fwd_plane_kernel ------------
// dY goes through the wave-private tile unchanged (the correlation with the flipped kernel), the producer's raw output XRAW of the
// same 4 pixels - for act' and the BatchNorm-backward sums - is prefetched beside it.
template <int K, int PRO, int W, int R = W>
__global__ void __launch_bounds__(NTHREADS) dwconv_dgrad_plane_kernel(const DwP p, int bchunk) {
    constexpr int PADK = (K - 1) / 2;
    constexpr bool BAND = R != W;
    constexpr int NB = W / R, GPP = R * W / 4, NG = GPP > 64 ? GPP / 64 : 1, LPP = GPP > 64 ? 64 : GPP, PW = 64 / LPP, XGW = W / 4;
    constexpr int HG = BAND ? (K - 1) * XGW : 0;
    constexpr int TH = R + K - 1, TW = W + 8, TILE = TH * TW, HW = W * W;
    static_assert(HG <= 64 && (!BAND || PW == 1) && W % R == 0, "band geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][2][PW][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= p.C) return;
    const int n_items = p.B * NB;
    const int i_lo = blockIdx.y * bchunk, i_hi = min(n_items, i_lo + bchunk);
    float scale = 1.0f, shift = 0.0f, mean = 0.0f, invstd = 1.0f;
    if (PRO != S2K_PRO_NONE) { scale = p.bnv[c]; shift = p.bnv[p.C + c]; mean = p.bnv[2 * p.C + c]; invstd = p.bnv[3 * p.C + c]; }
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = p.w[c * (K * K) + (K * K - 1 - i)];        // flipped
    float* tiles = smem + wave * (2 * PW * TILE);
    for (int i = lane; i < 2 * PW * TILE; i += 64) tiles[i] = 0.0f;
    const int sub = lane / LPP, li = lane % LPP;
    const int64_t bstride = (int64_t)p.C * HW, coff = (int64_t)c * HW;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int hrow = lane / XGW, hxg = lane % XGW;
    const int h_trow = hrow < PADK ? hrow : R + hrow;
    const int h_dy = hrow < PADK ? hrow - PADK : R + hrow - PADK;
    auto item_off = [&](int it, int& y0) -> int64_t {
        const int b = BAND ? it / NB : it;
        y0 = BAND ? (it - b * NB) * R : 0;
        return (int64_t)b * bstride + coff + (int64_t)y0 * W;
    };
    f32x4 cur[NG], xcur[NG], hcur = zero;
    {
        int y0;
        const int it = i_lo + sub;
        const bool ok0 = it < i_hi;
        const int64_t off = item_off(ok0 ? it : i_lo, y0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cur[g] = ok0 ? *reinterpret_cast<const f32x4*>(p.dy + off + 4 * li + 256 * g) : zero;
            xcur[g] = (ok0 && PRO != S2K_PRO_NONE) ? *reinterpret_cast<const f32x4*>(p.x + off + 4 * li + 256 * g) : zero;
        }
        if (BAND && lane < HG && ok0 && y0 + h_dy >= 0 && y0 + h_dy < W) hcur = *reinterpret_cast<const f32x4*>(p.dy + off + (int64_t)h_dy * W + 4 * hxg);
    }
    float s1 = 0.0f, s2 = 0.0f;
    int buf = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PW, buf ^= 1) {
        const int it = i0 + sub;
        const bool ok = it < i_hi, more = it + PW < i_hi;
        int y0, y1;
        const int64_t off = item_off(ok ? it : i_lo, y0);
        const int64_t offn = item_off(more ? it + PW : i_lo, y1);
        f32x4 nxt[NG], xnxt[NG], ob[NG], hnxt = zero;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            nxt[g] = more ? *reinterpret_cast<const f32x4*>(p.dy + offn + 4 * li + 256 * g) : zero;
            xnxt[g] = (more && PRO != S2K_PRO_NONE) ? *reinterpret_cast<const f32x4*>(p.x + offn + 4 * li + 256 * g) : zero;
            ob[g] = (p.beta && ok) ? *reinterpret_cast<const f32x4*>(p.out + off + 4 * li + 256 * g) : zero;
        }
        if (BAND && lane < HG && more && y1 + h_dy >= 0 && y1 + h_dy < W) hnxt = *reinterpret_cast<const f32x4*>(p.dy + offn + (int64_t)h_dy * W + 4 * hxg);
        float* t = tiles + (buf * PW + sub) * TILE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            *reinterpret_cast<f32x4*>(t + (r + PADK) * TW + 4 + 4 * xg) = ok ? cur[g] : zero;
        }
        if (BAND && lane < HG) *reinterpret_cast<f32x4*>(t + h_trow * TW + 4 + 4 * hxg) = (ok && y0 + h_dy >= 0 && y0 + h_dy < W) ? hcur : zero;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gi = li + 64 * g, r = gi / XGW, xg = gi % XGW;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            const float* t0 = t + r * TW + 4 * xg;
#pragma unroll
            for (int a = 0; a < K; ++a) {
                float win[12];
                read_window<3>(t0 + a * TW, win);
#pragma unroll
                for (int b = 0; b < K; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(wk[a * K + b], win[4 - PADK + j + b], o[j]);
            }
            if (ok) {
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] *= act_grad(fmaf(xcur[g][j], scale, shift), PRO);
                        s1 += o[j];
                        s2 = fmaf(o[j], (xcur[g][j] - mean) * invstd, s2);
                    }
                }
                *reinterpret_cast<f32x4*>(p.out + off + 4 * li + 256 * g) = f32x4{o[0] + ob[g][0], o[1] + ob[g][1], o[2] + ob[g][2], o[3] + ob[g][3]};
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) { cur[g] = nxt[g]; xcur[g] = xnxt[g]; }
        hcur = hnxt;
    }
    if (p.stats) {
        const double sd = wave_sum_d((double)s1), qd = wave_sum_d((double)s2);
        if (lane == 0) {
            double* st = p.stats + (int64_t)((blockIdx.x + blockIdx.y) % p.nrep) * 2 * p.C;
            atomic_add_d(st + c, sd);
            atomic_add_d(st + p.C + c, qd);
        }
    }
}

// ---- data gradient at stride 2 on even square planes: the wave-per-channel scheme --------------------------------------------------
// A lane owns one 4-pixel group of dY (row r, columns 4 xg ..) and computes the 2 x 8 input pixels below it (rows 2r, 2r + 1, columns
// 8 xg .. 8 xg + 7): input pixel (2r + py, 8 xg + 2q + px) only meets the taps with ky = (py + PT) mod 2, kx = (px + PL) mod 2 - every tap
// feeds exactly one of the four parities, and which dY element it reads ((py + PT - ky) / 2 rows, (px + PL - kx) / 2 columns away) is
// a compile-time constant: K * K multiply-adds per dY pixel, three aligned 16-byte LDS reads per dY row.
template <int K, int PRO, int WO, int RO>
__global__ void __launch_bounds__(NTHREADS) dwconv_dgrad_plane_s2_kernel(const DwP p, int bchunk) {
    constexpr int WI = 2 * WO, PT = (K - 2) / 2;
    constexpr int NB = WO / RO;
    constexpr bool BAND = NB > 1;
    constexpr int GO = RO * WO / 4, LPP = GO, PW = 64 / LPP, XGO = WO / 4;
    constexpr int HT = 1, HB = K == 5 ? 1 : 0;     // dY halo rows above / below a band (row offsets -1 .. +1 for k = 5, -1 .. 0 for k = 3)
    constexpr int HGT = BAND ? (HT + HB) * XGO : 0;
    constexpr int TH = RO + HT + HB, TW = WO + 8, TILE = TH * TW;
    constexpr int HWI = WI * WI, HWO = WO * WO;
    static_assert((GO == 64 || GO == 16) && HGT <= 64 && (!BAND || PW == 1), "stride-2 plane geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][2][PW][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= p.C) return;
    const int n_items = p.B * NB;
    const int i_lo = blockIdx.y * bchunk, i_hi = min(n_items, i_lo + bchunk);
    float scale = 1.0f, shift = 0.0f, mean = 0.0f, invstd = 1.0f;
    if (PRO != S2K_PRO_NONE) { scale = p.bnv[c]; shift = p.bnv[p.C + c]; mean = p.bnv[2 * p.C + c]; invstd = p.bnv[3 * p.C + c]; }
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = p.w[c * (K * K) + i];
    float* tiles = smem + wave * (2 * PW * TILE);
    for (int i = lane; i < 2 * PW * TILE; i += 64) tiles[i] = 0.0f;
    const int sub = lane / LPP, li = lane % LPP;
    const int r = li / XGO, xg = li % XGO;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int64_t istride = (int64_t)p.C * HWI, ostride = (int64_t)p.C * HWO;
    auto split = [&](int it, int& b, int& yo0) { b = BAND ? it / NB : it; yo0 = BAND ? (it - b * NB) * RO : 0; };
    const int hr = lane / XGO, hxg = lane % XGO;                        // halo group of this lane (lane < HGT)
    const int h_trow = hr < HT ? hr : RO + hr, h_dy = hr < HT ? hr - HT : RO + hr - HT;
    f32x4 cur = zero, hcur = zero, xcur[4];
    auto fetch = [&](int it, bool valid, f32x4& v, f32x4& hv, f32x4 (&xv)[4]) {
        int b, yo0;
        split(valid ? it : i_lo, b, yo0);
        const float* src = p.dy + (int64_t)b * ostride + (int64_t)c * HWO + (int64_t)yo0 * WO;
        v = valid ? *reinterpret_cast<const f32x4*>(src + 4 * li) : zero;
        hv = zero;
        if (BAND && lane < HGT && valid && yo0 + h_dy >= 0 && yo0 + h_dy < WO) hv = *reinterpret_cast<const f32x4*>(src + (int64_t)h_dy * WO + 4 * hxg);
        const float* xs = p.x + (int64_t)b * istride + (int64_t)c * HWI + (int64_t)(2 * (yo0 + r)) * WI + 8 * xg;
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = (valid && PRO != S2K_PRO_NONE) ? *reinterpret_cast<const f32x4*>(xs + (k >> 1) * WI + 4 * (k & 1)) : zero;
    };
    fetch(i_lo + sub, i_lo + sub < i_hi, cur, hcur, xcur);
    float s1 = 0.0f, s2 = 0.0f;
    int buf = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PW, buf ^= 1) {
        const int it = i0 + sub;
        const bool ok = it < i_hi;
        f32x4 nxt, hnxt, xnxt[4];
        fetch(it + PW, it + PW < i_hi, nxt, hnxt, xnxt);
        int b, yo0;
        split(ok ? it : i_lo, b, yo0);
        float* gout = p.out + (int64_t)b * istride + (int64_t)c * HWI + (int64_t)(2 * (yo0 + r)) * WI + 8 * xg;
        f32x4 ob[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ob[k] = (p.beta && ok) ? *reinterpret_cast<const f32x4*>(gout + (k >> 1) * WI + 4 * (k & 1)) : zero;
        float* t = tiles + (buf * PW + sub) * TILE;
        *reinterpret_cast<f32x4*>(t + (r + HT) * TW + 4 + 4 * xg) = ok ? cur : zero;
        if (BAND && lane < HGT) *reinterpret_cast<f32x4*>(t + h_trow * TW + 4 + 4 * hxg) = (ok && yo0 + h_dy >= 0 && yo0 + h_dy < WO) ? hcur : zero;
        __builtin_amdgcn_wave_barrier();
        float acc[2][4][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][q][0] = acc[a][q][1] = 0.0f;
        float win[3][12];
#pragma unroll
        for (int d = 0; d < HT + 1 + HB; ++d) read_window<3>(t + (r + d) * TW + 4 * xg, win[d]);      // dY rows r - 1 .. r (+ 1), dY columns 4 xg - 4 .. 4 xg + 7
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int py = (ky + PT) & 1, dyo = (py + PT - ky) >> 1;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int px = (kx + PT) & 1, dxo = (px + PT - kx) >> 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[py][q][px] = fmaf(wk[ky * K + kx], win[dyo + HT][4 + q + dxo], acc[py][q][px]);
            }
        }
        if (ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int py = k >> 1, h = k & 1;                       // row 2r + py, columns 8 xg + 4 h .. + 3
                float o[4] = {acc[py][2 * h][0], acc[py][2 * h][1], acc[py][2 * h + 1][0], acc[py][2 * h + 1][1]};
                if (PRO != S2K_PRO_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] *= act_grad(fmaf(xcur[k][j], scale, shift), PRO);
                        s1 += o[j];
                        s2 = fmaf(o[j], (xcur[k][j] - mean) * invstd, s2);
                    }
                }
                *reinterpret_cast<f32x4*>(gout + py * WI + 4 * h) = f32x4{o[0] + ob[k][0], o[1] + ob[k][1], o[2] + ob[k][2], o[3] + ob[k][3]};
            }
        }
        cur = nxt; hcur = hnxt;
#pragma unroll
        for (int k = 0; k < 4; ++k) xcur[k] = xnxt[k];
    }
    if (p.stats) {
        const double sd = wave_sum_d((double)s1), qd = wave_sum_d((double)s2);
        if (lane == 0) {
            double* st = p.stats + (int64_t)((blockIdx.x + blockIdx.y) % p.nrep) * 2 * p.C;
            atomic_add_d(st + c, sd);
            atomic_add_d(st + p.C + c, qd);
        }
    }
}

// ---- data gradient, stride 2 (4 layers of a b5) ------------------------------------------------------
// A lane computes a 2x2 block of input pixels.  Input pixel (iy, ix) only meets the taps with (iy + PT - ky) and
// (ix + PL - kx) even, i.e. each of the K*K taps feeds exactly ONE of the block's four pixels, and which one depends only on
// the parities QY = (iy0 + PT) & 1, QX = PL & 1 of the band: the kernel body is instantiated for the four (QY, QX) and every
// tap -> pixel assignment and tile offset is a compile-time constant (K*K multiply-adds per block; the per-pixel gather it
// replaces walked all K*K taps for every pixel and skipped three quarters of them behind a parity branch).
// Tile: dY rows r0 .. r0 + IRt - 1 (r0 = floor((iy0 + PT - K + 1) / 2), may be negative), columns -1 .. LW - 2, zeros outside.
template <int K, int QY, int QX>
__device__ __forceinline__ void dgrad_s2_block(const float* tp, int LW, const float* wk, float (&acc)[2][2]) {
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int dyy = QY ^ (ky & 1), dxx = QX ^ (kx & 1);       // the block pixel this tap feeds
            // row / column of dY relative to the lane's base (a + A0, b + B0): (QY + dyy - ky) / 2 and (QX + dxx - kx) / 2
            const int cy = (QY + dyy - ky) >> 1, cx = (QX + dxx - kx) >> 1;   // arithmetic shift: exact, the sums are even
            acc[dyy][dxx] = fmaf(wk[ky * K + kx], tp[cy * LW + cx], acc[dyy][dxx]);
        }
}

template <int K>
__global__ void __launch_bounds__(NTHREADS) dwconv_dgrad_s2_kernel(const DwP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [PPB][IRt][LW]  (dY rows, zero outside)
    float* wsm = smem + p.PPB * p.IRt * p.LW;         // [PPB][K*K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int K2 = K * K;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int iy0 = band * p.RT;                 // first input row of the band
    const int rows = min(p.RT, p.H - iy0);
    const int e = iy0 + p.PT, qy = e & 1, qx = p.PL & 1;
    const int lo = e - (K - 1);
    const int r0 = lo >= 0 ? lo >> 1 : -((1 - lo) >> 1);          // floor(lo / 2)
    stage_band<S2K_PRO_NONE>(p, p.dy, p.HO, p.WO, tile, pl0, nplanes, p.IRt, r0, -1);
    for (int idx = tid; idx < p.PPB * K2; idx += NTHREADS) {
        const int pl = idx / K2;
        const int64_t plane = pl0 + pl;
        wsm[idx] = p.w[chan_of(plane < nplanes ? plane : 0, p.C) * K2 + (idx - pl * K2)];
    }
    __syncthreads();
    const int A0 = ((e - qy) >> 1) - r0, B0 = ((p.PL - qx) >> 1) + 1;   // tile row / column of block (0, 0)'s reference tap
    const int bw = (p.W + 1) >> 1, bh = (rows + 1) >> 1;               // 2x2 blocks of the band
    const int n_blk = bw * bh;
    const int chunks_per_plane = (n_blk + 63) >> 6;
    const int wpp = p.PPB >= 4 ? 1 : 4 / p.PPB;
    const int per_plane = p.IRt * p.LW;
    double* st = p.stats ? p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.C : nullptr;
    for (int pl = wave / wpp; pl < p.PPB; pl += 4 / wpp) {
        const int64_t plane = pl0 + pl;
        if (plane >= nplanes) break;
        const int c = chan_of(plane, p.C);
        float s1 = 0.0f, s2 = 0.0f;
        float scale = 1.0f, shift = 0.0f, mean = 0.0f, invstd = 1.0f;
        if (p.pro != S2K_PRO_NONE) { scale = p.bnv[c]; shift = p.bnv[p.C + c]; mean = p.bnv[2 * p.C + c]; invstd = p.bnv[3 * p.C + c]; }
        const float* wk = wsm + pl * K2;
        for (int chn = wave % wpp; chn < chunks_per_plane; chn += wpp) {
            const int o = chn * 64 + lane;
            if (o < n_blk) {
                const int a = o / bw, b = o - a * bw;
                const float* tp = tile + pl * per_plane + (a + A0) * p.LW + b + B0;
                float acc[2][2];
                if (qy == 0 && qx == 0) dgrad_s2_block<K, 0, 0>(tp, p.LW, wk, acc);
                else if (qy == 0) dgrad_s2_block<K, 0, 1>(tp, p.LW, wk, acc);
                else if (qx == 0) dgrad_s2_block<K, 1, 0>(tp, p.LW, wk, acc);
                else dgrad_s2_block<K, 1, 1>(tp, p.LW, wk, acc);
#pragma unroll
                for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
                    for (int dxx = 0; dxx < 2; ++dxx) {
                        const int r = 2 * a + dyy, ix = 2 * b + dxx;
                        if (r < rows && ix < p.W) {
                            const int64_t off = plane * p.H * p.W + (int64_t)(iy0 + r) * p.W + ix;
                            float v = acc[dyy][dxx];
                            if (p.pro != S2K_PRO_NONE) {
                                const float xr = p.x[off];
                                v *= act_grad(fmaf(xr, scale, shift), p.pro);
                                s1 += v;
                                s2 = fmaf(v, (xr - mean) * invstd, s2);
                            }
                            if (p.beta) v += p.out[off];
                            p.out[off] = v;
                        }
                    }
            }
        }
        if (st) {
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {
                atomic_add_d(st + c, (double)s1);
                atomic_add_d(st + p.C + c, (double)s2);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
template <typename T>
static T* ref_ptr(const Ctx& c, int64_t ref) {
    if (ref < 0) return nullptr;
    const int base = (int)(ref >> 56);
    const int64_t off = ref & ((1ll << 56) - 1);
    if (base >= c.n_bases || c.bases[base] == nullptr) return reinterpret_cast<T*>(1);
    return reinterpret_cast<T*>(static_cast<char*>(c.bases[base]) + off);
}

static bool bad(const void* q) { return q == reinterpret_cast<const void*>(1); }

static int fill_geo(DwP& p, const int32_t* d) {
    p.B = d[0]; p.C = d[1]; p.H = d[2]; p.W = d[3]; p.K = d[4]; p.S = d[5]; p.PT = d[6]; p.PL = d[7];
    p.HO = d[8]; p.WO = d[9]; p.pro = d[10]; p.nrep = 1; p.beta = 0;
    if (p.B <= 0 || p.C <= 0 || p.H <= 0 || p.W <= 0 || (p.K != 3 && p.K != 5) || (p.S != 1 && p.S != 2)) {
        set_error("dwconv: unsupported geometry K=%d S=%d", p.K, p.S);
        return S2K_EINVAL;
    }
    if ((int64_t)p.B * p.C * p.H * p.W * 4 >= 0x7ffffff0ll) { set_error("dwconv: tensor larger than 2 GiB"); return S2K_EINVAL; }
    if (p.pro != S2K_PRO_NONE && p.pro != S2K_PRO_SILU) { set_error("dwconv: only the SiLU prologue is on this path"); return S2K_EINVAL; }
    return S2K_OK;
}

static int pow2ceil(int v) { int r = 1; while (r < v) r <<= 1; return r; }

// tiling over rows of the COMPUTED plane (ho x wo); src rows needed per band = irt, src columns = lw
static size_t tile_rows(DwP& p, int ho, int wo, int (*irt_for_rt)(int, const DwP&), int lw, bool with_w, bool vec) {
    // outputs per workgroup: enough loads in flight per CU to cover HBM latency, LDS small enough for ~6 workgroups / CU
    static const int target0 = tune_int("S2K_DW_TARGET", 4096);
    const int hw = ho * wo;
    p.LW = vec ? (lw + 3) & ~3 : lw | 1;     // vector layout: float4 rows; scalar layout: odd stride
    p.XG = cdiv(wo, 4);
    p.inv_xg = 1.0f / (float)p.XG;
    size_t lds = 0;
    for (int target = target0;; target >>= 1) {
        if (hw <= target) { p.PPB = target / hw; if (p.PPB > 32) p.PPB = 32; p.RT = ho; }
        else { p.PPB = 1; p.RT = target / wo; if (p.RT < 1) p.RT = 1; }
        p.IRt = irt_for_rt(p.RT, p);
        lds = ((size_t)p.PPB * p.IRt * p.LW + (with_w ? p.PPB * p.K * p.K : 4 * p.K * p.K)) * sizeof(float);
        if (lds <= 26 * 1024 || (p.PPB == 1 && p.RT == 1)) break;
    }
    p.bands = cdiv(ho, p.RT);
    const int items = p.RT * p.XG;
    p.LPP = items >= 64 ? 64 : pow2ceil(items);
    int l2 = 0;
    while ((1 << l2) < p.LW && l2 < 6) ++l2;
    p.lwp_shift = l2;
    l2 = 0;
    while ((1 << l2) < (p.LW >> 2) && l2 < 6) ++l2;
    p.vps_shift = l2;
    return lds;
}

static int irt_fwd(int rt, const DwP& p) { return (rt - 1) * p.S + p.K; }
static int irt_dgrad1(int rt, const DwP& p) { return rt + p.K - 1; }
static int irt_dgrad2(int rt, const DwP& p) { return (rt + p.K - 2) / 2 + 3; }   // dY rows a band of rt input rows reaches (+ slack for the floor / 2x2 blocks)

template <typename KernT>
static int launch_dw(KernT kern, const DwP& p, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) { set_error("dwconv: tile too large (%zu B)", lds); return S2K_EINVAL; }
    const int64_t groups = cdiv64((int64_t)p.B * p.C, p.PPB);
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * p.bands)), dim3(NTHREADS), lds, st, p);
    return S2K_OK;
}

int launch_dwconv_fwd(const S2kOp& op, const Ctx& c) {
    DwP p{};
    if (int e = fill_geo(p, op.d)) return e;
    p.x = ref_ptr<const float>(c, op.t[S2K_DWCONV_FWD_T_X]);
    p.bnv = ref_ptr<const float>(c, op.t[S2K_DWCONV_FWD_T_BNV]);
    p.w = ref_ptr<const float>(c, op.t[S2K_DWCONV_FWD_T_WT]);
    p.out = ref_ptr<float>(c, op.t[S2K_DWCONV_FWD_T_Y]);
    p.stats = ref_ptr<double>(c, op.t[S2K_DWCONV_FWD_T_STATS]);
    if (op.d[S2K_DWCONV_FWD_D_NREP] > 0) p.nrep = op.d[S2K_DWCONV_FWD_D_NREP];
    if (bad(p.x) || bad(p.bnv) || bad(p.w) || bad(p.out) || bad(p.stats)) { set_error("dwconv_fwd: null base"); return S2K_EFAULT; }
    if (!p.x || !p.w || !p.out || (p.pro != S2K_PRO_NONE && !p.bnv)) { set_error("dwconv_fwd: missing tensor"); return S2K_EINVAL; }
    if (int e = fill_bn_fold(p.fold, c, &op.t[S2K_DWCONV_FWD_T_FSTATS], op.n[S2K_DWCONV_FWD_N_FCOUNT], op.d[S2K_DWCONV_FWD_D_FNREP],
                             op.f[S2K_DWCONV_FWD_F_FEPS], op.f[S2K_DWCONV_FWD_F_FMOM], const_cast<float*>(p.bnv), "dwconv_fwd")) return e;
    if (p.fold.stats && p.pro == S2K_PRO_NONE) { set_error("dwconv_fwd: FSTATS without a prologue"); return S2K_EINVAL; }
    const bool silu = p.pro == S2K_PRO_SILU;
    static const int plane_on = tune_int("S2K_DW_PLANE", 1);
    static const int band_on = tune_int("S2K_DW_BAND", 1);
    if (plane_on && p.S == 1 && p.H == p.W && (p.W == 8 || p.W == 16 || p.W == 32 || (band_on && (p.W == 64 || (p.W == 128 && p.K == 3)))) &&
        p.HO == p.H && p.WO == p.W && (p.K == 3 || p.K == 5) && p.PT == (p.K - 1) / 2 && p.PL == (p.K - 1) / 2 && (p.pro == S2K_PRO_NONE || silu)) {
        // square planes at stride 1: one wave per channel walks (image, band) items with no workgroup barrier (dwconv_fwd_plane_kernel)
        const int pw = p.W == 8 ? 4 : 1;
        const int rows = p.W == 64 ? 16 : p.W == 128 ? 8 : p.W;              // rows per work item
        const int n_items = p.B * (p.W / rows);
        static const int plane_waves = tune_int("S2K_DW_PLANE_WAVES", 6144);
        int bsplit = std::max(1, std::min(cdiv(n_items, 2 * pw), cdiv(plane_waves, p.C)));      // ~24 waves per CU, at least two passes per wave
        const int bchunk = cdiv(cdiv(n_items, bsplit), pw) * pw;
        bsplit = cdiv(n_items, bchunk);
        const size_t lds = (size_t)4 * 2 * pw * (rows + p.K - 1) * (p.W + 8) * sizeof(float);
        const dim3 grid(cdiv(p.C, 4), bsplit);
#define DW_PLANE(KK, WW, RR) do { \
            if (silu) hipLaunchKernelGGL((dwconv_fwd_plane_kernel<KK, S2K_PRO_SILU, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); \
            else hipLaunchKernelGGL((dwconv_fwd_plane_kernel<KK, S2K_PRO_NONE, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); } while (0)
        if (p.K == 3) {
            if (p.W == 8) DW_PLANE(3, 8, 8); else if (p.W == 16) DW_PLANE(3, 16, 16); else if (p.W == 32) DW_PLANE(3, 32, 32);
            else if (p.W == 64) DW_PLANE(3, 64, 16); else DW_PLANE(3, 128, 8);
        } else {
            if (p.W == 8) DW_PLANE(5, 8, 8); else if (p.W == 16) DW_PLANE(5, 16, 16); else if (p.W == 32) DW_PLANE(5, 32, 32);
            else DW_PLANE(5, 64, 16);
        }
#undef DW_PLANE
        return S2K_OK;
    }
    static const int s2_on = tune_int("S2K_DW_PLANE_S2", 1);
    if (plane_on && s2_on && p.S == 2 && p.H == p.W && p.HO == p.WO && p.H == 2 * p.HO && (p.WO == 8 || p.WO == 16 || p.WO == 32 || p.WO == 64) &&
        (p.K == 3 || p.K == 5) && p.PT == (p.K - 2) / 2 && p.PL == (p.K - 2) / 2 && (p.pro == S2K_PRO_NONE || silu)) {
        // stride 2 on even square planes (dwconv_fwd_plane_s2_kernel)
        const int pw = p.WO == 8 ? 4 : 1;
        const int ro = p.WO == 64 ? 4 : p.WO == 32 ? 8 : p.WO;              // output rows per work item
        const int n_items = p.B * (p.WO / ro);
        static const int plane_waves = tune_int("S2K_DW_PLANE_WAVES", 6144);
        int bsplit = std::max(1, std::min(cdiv(n_items, 2 * pw), cdiv(plane_waves, p.C)));
        const int bchunk = cdiv(cdiv(n_items, bsplit), pw) * pw;
        bsplit = cdiv(n_items, bchunk);
        const size_t lds = (size_t)4 * 2 * pw * (2 * ro + p.K - 2) * (p.W + 8) * sizeof(float);
        const dim3 grid(cdiv(p.C, 4), bsplit);
#define DW_S2(KK, WW, RR) do { \
            if (silu) hipLaunchKernelGGL((dwconv_fwd_plane_s2_kernel<KK, S2K_PRO_SILU, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); \
            else hipLaunchKernelGGL((dwconv_fwd_plane_s2_kernel<KK, S2K_PRO_NONE, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); } while (0)
        if (p.K == 3) {
            if (p.WO == 8) DW_S2(3, 8, 8); else if (p.WO == 16) DW_S2(3, 16, 16); else if (p.WO == 32) DW_S2(3, 32, 8); else DW_S2(3, 64, 4);
        } else {
            if (p.WO == 8) DW_S2(5, 8, 8); else if (p.WO == 16) DW_S2(5, 16, 16); else if (p.WO == 32) DW_S2(5, 32, 8); else DW_S2(5, 64, 4);
        }
#undef DW_S2
        return S2K_OK;
    }
    // tile columns: source column cc - 4; the widest window ends at 4 - PL + (4*XG - 1)*S + K - 1
    const int lw = 4 - p.PL + (cdiv(p.WO, 4) * 4 - 1) * p.S + p.K;
    if (p.PL < 0 || p.PL > 2) { set_error("dwconv: left padding %d is not on this path", p.PL); return S2K_EINVAL; }
    const size_t lds = tile_rows(p, p.HO, p.WO, irt_fwd, lw, true, true) + (p.fold.stats ? 2 * (size_t)p.PPB * sizeof(float) : 0);
#define DW_FWD(KK, SS, PP) (silu ? launch_dw(dwconv_fwd_kernel<KK, SS, PP, S2K_PRO_SILU>, p, lds, c.stream) \
                                 : launch_dw(dwconv_fwd_kernel<KK, SS, PP, S2K_PRO_NONE>, p, lds, c.stream))
#define DW_FWD_PL(KK, SS) (p.PL == 0 ? DW_FWD(KK, SS, 0) : p.PL == 1 ? DW_FWD(KK, SS, 1) : DW_FWD(KK, SS, 2))
    if (p.K == 3 && p.S == 1) return DW_FWD_PL(3, 1);
    if (p.K == 3 && p.S == 2) return DW_FWD_PL(3, 2);
    if (p.K == 5 && p.S == 1) return DW_FWD_PL(5, 1);
    return DW_FWD_PL(5, 2);
#undef DW_FWD_PL
#undef DW_FWD
}

int launch_dwconv_wgrad(const S2kOp& op, const Ctx& c) {
    DwP p{};
    if (int e = fill_geo(p, op.d)) return e;
    p.dy = ref_ptr<const float>(c, op.t[S2K_DWCONV_WGRAD_T_DY]);
    p.x = ref_ptr<const float>(c, op.t[S2K_DWCONV_WGRAD_T_X]);
    p.bnv = ref_ptr<const float>(c, op.t[S2K_DWCONV_WGRAD_T_BNV]);
    p.out = ref_ptr<float>(c, op.t[S2K_DWCONV_WGRAD_T_DW]);
    if (bad(p.x) || bad(p.bnv) || bad(p.dy) || bad(p.out)) { set_error("dwconv_wgrad: null base"); return S2K_EFAULT; }
    if (!p.x || !p.dy || !p.out || (p.pro != S2K_PRO_NONE && !p.bnv)) { set_error("dwconv_wgrad: missing tensor"); return S2K_EINVAL; }
    static const int plane_on = tune_int("S2K_DW_PLANE", 1);
    if (plane_on && p.S == 1 && p.H == p.W && (p.W == 8 || p.W == 16 || p.W == 32 || p.W == 64 || (p.W == 128 && p.K == 3)) &&
        p.HO == p.H && p.WO == p.W && (p.K == 3 || p.K == 5) && p.PT == (p.K - 1) / 2 && p.PL == (p.K - 1) / 2 &&
        (p.pro == S2K_PRO_NONE || p.pro == S2K_PRO_SILU)) {
        // square planes at stride 1: one wave per channel walks (image, band) items (dwconv_wgrad_plane_kernel)
        const int pw = p.W == 8 ? 4 : 1;
        const int rows = p.W == 64 ? 16 : p.W == 128 ? 8 : p.W;
        const int n_items = p.B * (p.W / rows);
        static const int plane_waves = tune_int("S2K_DW_PLANE_WAVES", 6144);
        int bsplit = std::max(1, std::min(cdiv(n_items, 2 * pw), cdiv(plane_waves, p.C)));
        // every wave ends in K * K float atomics on its channel's taps, and a 128-byte line of DW holds the taps of ~3 channels: atomics
        // on one line execute one after the other (24 channels at 128 x 128 with 256 waves per channel: 8 k atomics per line = 0.1 ms
        // for a 0.03 ms contraction), so at most 48 waves share a channel
        static const int wg_split_max = tune_int("S2K_DW_WGP_SPLIT_MAX", 48);
        bsplit = std::min(bsplit, wg_split_max);
        const int bchunk = cdiv(cdiv(n_items, bsplit), pw) * pw;
        bsplit = cdiv(n_items, bchunk);
        const size_t lds = (size_t)4 * 2 * pw * (rows + p.K - 1) * (p.W + 8) * sizeof(float);
        const dim3 grid(cdiv(p.C, 4), bsplit);
        const bool sl = p.pro == S2K_PRO_SILU;
#define DW_WGP(KK, WW, RR) do { \
            if (sl) hipLaunchKernelGGL((dwconv_wgrad_plane_kernel<KK, S2K_PRO_SILU, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); \
            else hipLaunchKernelGGL((dwconv_wgrad_plane_kernel<KK, S2K_PRO_NONE, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); } while (0)
        if (p.K == 3) {
            if (p.W == 8) DW_WGP(3, 8, 8); else if (p.W == 16) DW_WGP(3, 16, 16); else if (p.W == 32) DW_WGP(3, 32, 32);
            else if (p.W == 64) DW_WGP(3, 64, 16); else DW_WGP(3, 128, 8);
        } else {
            if (p.W == 8) DW_WGP(5, 8, 8); else if (p.W == 16) DW_WGP(5, 16, 16); else if (p.W == 32) DW_WGP(5, 32, 32);
            else DW_WGP(5, 64, 16);
        }
#undef DW_WGP
        return S2K_OK;
    }
    const int lw = 4 - p.PL + (cdiv(p.WO, 4) * 4 - 1) * p.S + p.K;
    if (p.PL < 0 || p.PL > 2) { set_error("dwconv: left padding %d is not on this path", p.PL); return S2K_EINVAL; }
    size_t lds = tile_rows(p, p.HO, p.WO, irt_fwd, lw, false, true);
    static const int bloop_on = tune_int("S2K_DW_WG_BLOOP", 1);
    if (bloop_on && p.bands == 1 && p.HO * p.WO <= 1024 && p.B > 1) {
        // small planes: channels per workgroup = 4 waves x (64 / LPP) planes per wave; images are walked inside the kernel
        const int ppb = 4 * (64 / p.LPP);
        const size_t lds2 = ((size_t)ppb * p.IRt * p.LW + (size_t)ppb * p.K * p.K) * sizeof(float);
        if (lds2 <= 64 * 1024) {
            p.PPB = ppb;
            lds = lds2;
            p.cgroups = cdiv(p.C, ppb);
            int bsplits = cdiv(768, p.cgroups);           // ~3 workgroups per CU
            if (bsplits > p.B) bsplits = p.B;
            if (bsplits < 1) bsplits = 1;
            p.bloop = cdiv(p.B, bsplits);
            bsplits = cdiv(p.B, p.bloop);
            const bool silu_b = p.pro == S2K_PRO_SILU;
            const dim3 grid((unsigned)(p.cgroups * bsplits));
#define DW_WGB(KK, SS, PP) do { if (silu_b) hipLaunchKernelGGL((dwconv_wgrad_kernel<KK, SS, PP, S2K_PRO_SILU>), grid, dim3(NTHREADS), lds, c.stream, p); \
                                else hipLaunchKernelGGL((dwconv_wgrad_kernel<KK, SS, PP, S2K_PRO_NONE>), grid, dim3(NTHREADS), lds, c.stream, p); return S2K_OK; } while (0)
#define DW_WGB_PL(KK, SS) do { if (p.PL == 0) DW_WGB(KK, SS, 0); else if (p.PL == 1) DW_WGB(KK, SS, 1); else DW_WGB(KK, SS, 2); } while (0)
            if (p.K == 3 && p.S == 1) DW_WGB_PL(3, 1);
            if (p.K == 3 && p.S == 2) DW_WGB_PL(3, 2);
            if (p.K == 5 && p.S == 1) DW_WGB_PL(5, 1);
            DW_WGB_PL(5, 2);
#undef DW_WGB_PL
#undef DW_WGB
        }
    }
    const bool silu = p.pro == S2K_PRO_SILU;
#define DW_WG(KK, SS, PP) (silu ? launch_dw(dwconv_wgrad_kernel<KK, SS, PP, S2K_PRO_SILU>, p, lds, c.stream) \
                                : launch_dw(dwconv_wgrad_kernel<KK, SS, PP, S2K_PRO_NONE>, p, lds, c.stream))
#define DW_WG_PL(KK, SS) (p.PL == 0 ? DW_WG(KK, SS, 0) : p.PL == 1 ? DW_WG(KK, SS, 1) : DW_WG(KK, SS, 2))
    if (p.K == 3 && p.S == 1) return DW_WG_PL(3, 1);
    if (p.K == 3 && p.S == 2) return DW_WG_PL(3, 2);
    if (p.K == 5 && p.S == 1) return DW_WG_PL(5, 1);
    return DW_WG_PL(5, 2);
#undef DW_WG_PL
#undef DW_WG
}

int launch_dwconv_dgrad(const S2kOp& op, const Ctx& c) {
    DwP p{};
    if (int e = fill_geo(p, op.d)) return e;
    p.beta = op.d[S2K_DWCONV_DGRAD_D_BETA];
    if (op.d[S2K_DWCONV_DGRAD_D_NREP] > 0) p.nrep = op.d[S2K_DWCONV_DGRAD_D_NREP];
    p.dy = ref_ptr<const float>(c, op.t[S2K_DWCONV_DGRAD_T_DY]);
    p.w = ref_ptr<const float>(c, op.t[S2K_DWCONV_DGRAD_T_WT]);
    p.x = ref_ptr<const float>(c, op.t[S2K_DWCONV_DGRAD_T_XRAW]);
    p.bnv = ref_ptr<const float>(c, op.t[S2K_DWCONV_DGRAD_T_BNV]);
    p.out = ref_ptr<float>(c, op.t[S2K_DWCONV_DGRAD_T_G]);
    p.stats = ref_ptr<double>(c, op.t[S2K_DWCONV_DGRAD_T_STATS2]);
    if (bad(p.x) || bad(p.bnv) || bad(p.dy) || bad(p.out) || bad(p.w) || bad(p.stats)) { set_error("dwconv_dgrad: null base"); return S2K_EFAULT; }
    if (!p.dy || !p.w || !p.out || (p.pro != S2K_PRO_NONE && (!p.bnv || !p.x))) { set_error("dwconv_dgrad: missing tensor"); return S2K_EINVAL; }
    if (p.pro == S2K_PRO_NONE) p.stats = nullptr;
    if (p.S == 1) {
        if (p.HO != p.H || p.WO != p.W) { set_error("dwconv_dgrad: stride-1 geometry mismatch"); return S2K_EINVAL; }
        const int pr = p.K - 1 - p.PL;       // right padding of the forward = left reach of the correlation
        if (pr < 0 || pr > 2) { set_error("dwconv_dgrad: padding %d is not on this path", p.PL); return S2K_EINVAL; }
        static const int plane_on = tune_int("S2K_DW_PLANE", 1);
        static const int band_on = tune_int("S2K_DW_BAND", 1);
        if (plane_on && p.H == p.W && (p.W == 8 || p.W == 16 || p.W == 32 || (band_on && (p.W == 64 || (p.W == 128 && p.K == 3)))) &&
            (p.K == 3 || p.K == 5) && p.PT == (p.K - 1) / 2 && p.PL == (p.K - 1) / 2 && (p.pro == S2K_PRO_NONE || p.pro == S2K_PRO_SILU)) {
            // square planes at stride 1: one wave per channel walks (image, band) items (dwconv_dgrad_plane_kernel)
            const int pw = p.W == 8 ? 4 : 1;
            const int rows = p.W == 64 ? 16 : p.W == 128 ? 8 : p.W;
            const int n_items = p.B * (p.W / rows);
            static const int plane_waves = tune_int("S2K_DW_PLANE_WAVES", 6144);
            int bsplit = std::max(1, std::min(cdiv(n_items, 2 * pw), cdiv(plane_waves, p.C)));
            const int bchunk = cdiv(cdiv(n_items, bsplit), pw) * pw;
            bsplit = cdiv(n_items, bchunk);
            const size_t lds = (size_t)4 * 2 * pw * (rows + p.K - 1) * (p.W + 8) * sizeof(float);
            const dim3 grid(cdiv(p.C, 4), bsplit);
            const bool sl = p.pro == S2K_PRO_SILU;
#define DW_DGP(KK, WW, RR) do { \
                if (sl) hipLaunchKernelGGL((dwconv_dgrad_plane_kernel<KK, S2K_PRO_SILU, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); \
                else hipLaunchKernelGGL((dwconv_dgrad_plane_kernel<KK, S2K_PRO_NONE, WW, RR>), grid, dim3(NTHREADS), lds, c.stream, p, bchunk); } while (0)
            if (p.K == 3) {
                if (p.W == 8) DW_DGP(3, 8, 8); else if (p.W == 16) DW_DGP(3, 16, 16); else if (p.W == 32) DW_DGP(3, 32, 32);
                else if (p.W == 64) DW_DGP(3, 64, 16); else DW_DGP(3, 128, 8);
            } else {
                if (p.W == 8) DW_DGP(5, 8, 8); else if (p.W == 16) DW_DGP(5, 16, 16); else if (p.W == 32) DW_DGP(5, 32, 32);
                else DW_DGP(5, 64, 16);
            }
#undef DW_DGP
            return S2K_OK;
        }
        const int lw = 4 - pr + cdiv(p.W, 4) * 4 + p.K - 1;
        const size_t lds = tile_rows(p, p.H, p.W, irt_dgrad1, lw, true, true);
        const bool silu = p.pro == S2K_PRO_SILU;
#define DW_DG(KK, PP) (silu ? launch_dw(dwconv_dgrad_s1_kernel<KK, PP, S2K_PRO_SILU>, p, lds, c.stream) \
                            : launch_dw(dwconv_dgrad_s1_kernel<KK, PP, S2K_PRO_NONE>, p, lds, c.stream))
        if (p.K == 3) return pr == 0 ? DW_DG(3, 0) : pr == 1 ? DW_DG(3, 1) : DW_DG(3, 2);
        return pr == 0 ? DW_DG(5, 0) : pr == 1 ? DW_DG(5, 1) : DW_DG(5, 2);
#undef DW_DG
    }
    static const int s2_on = tune_int("S2K_DW_PLANE_S2", 1);
    if (s2_on && p.S == 2 && p.H == p.W && p.HO == p.WO && p.H == 2 * p.HO && (p.WO == 8 || p.WO == 16 || p.WO == 32 || p.WO == 64) &&
        (p.K == 3 || p.K == 5) && p.PT == (p.K - 2) / 2 && p.PL == (p.K - 2) / 2 && (p.pro == S2K_PRO_NONE || p.pro == S2K_PRO_SILU)) {
        // even square planes (dwconv_dgrad_plane_s2_kernel)
        const int pw = p.WO == 8 ? 4 : 1;
        const int ro = p.WO == 64 ? 4 : p.WO == 32 ? 8 : p.WO;              // dY rows per work item
        const int n_items = p.B * (p.WO / ro);
        static const int plane_waves = tune_int("S2K_DW_PLANE_WAVES", 6144);
        int bsplit = std::max(1, std::min(cdiv(n_items, 2 * pw), cdiv(plane_waves, p.C)));
        const int bchunk = cdiv(cdiv(n_items, bsplit), pw) * pw;
        bsplit = cdiv(n_items, bchunk);
        const size_t lds2 = (size_t)4 * 2 * pw * (ro + (p.K == 5 ? 2 : 1)) * (p.WO + 8) * sizeof(float);
        const dim3 grid(cdiv(p.C, 4), bsplit);
        const bool sl = p.pro == S2K_PRO_SILU;
#define DW_DG2(KK, WW, RR) do { \
            if (sl) hipLaunchKernelGGL((dwconv_dgrad_plane_s2_kernel<KK, S2K_PRO_SILU, WW, RR>), grid, dim3(NTHREADS), lds2, c.stream, p, bchunk); \
            else hipLaunchKernelGGL((dwconv_dgrad_plane_s2_kernel<KK, S2K_PRO_NONE, WW, RR>), grid, dim3(NTHREADS), lds2, c.stream, p, bchunk); } while (0)
        if (p.K == 3) {
            if (p.WO == 8) DW_DG2(3, 8, 8); else if (p.WO == 16) DW_DG2(3, 16, 16); else if (p.WO == 32) DW_DG2(3, 32, 8); else DW_DG2(3, 64, 4);
        } else {
            if (p.WO == 8) DW_DG2(5, 8, 8); else if (p.WO == 16) DW_DG2(5, 16, 16); else if (p.WO == 32) DW_DG2(5, 32, 8); else DW_DG2(5, 64, 4);
        }
#undef DW_DG2
        return S2K_OK;
    }
    const size_t lds = tile_rows(p, p.H, p.W, irt_dgrad2, p.WO + 2, true, false);   // one zero column on either side
    return p.K == 3 ? launch_dw(dwconv_dgrad_s2_kernel<3>, p, lds, c.stream) : launch_dw(dwconv_dgrad_s2_kernel<5>, p, lds, c.stream);
}

}  // namespace s2k
