// Implicit-GEMM convolution on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16), f32 accumulate: the bf16-MIXED mode
// (reported separately from the f32 parity path; the reference itself trains with precision="bf16",
// /root/reference/src/configs/segmentation.py:146,153).  Same stage record, same operand tensors (raw f32 NCHW activations with
// the BatchNorm + activation (+ SE gate) prologue applied on load, channel concat as two source pointers), same f32 epilogue
// (bias, residual, accumulate, BatchNorm batch statistics in f64) as igemm.hip / igemm_pc.hip; only the two MFMA operands are
// rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32): weights once per step by WEIGHT_PACK, activations when a tile is
// written to LDS.  Replaces the same reference code (efficientnet_unet.py:168-176,288-297,319-372 under autocast).
//
// Why the structure differs from the f32 kernels: a f32 MFMA occupies its SIMD's issue port for its whole 64 cycles, so the
// f32 kernels split waves into MFMA-only consumers and loaders.  The bf16 MFMA does 8x the work per instruction and blocks
// vector issue for 8 of its 32 cycles: every wave can load, convert and multiply.  What bounds these kernels is HBM / L2
// bandwidth (the activations are still f32 in memory), so:
//   * 4 waves per workgroup (2 x 2 or 1 x 4 over the output tile), 2-3 workgroups per CU hide each other's barriers;
//   * ONE LDS image, register-staged prefetch: the next K chunk's global loads are issued before the MFMAs of the current
//     chunk and written (prologue, bf16 rounding, transposition) after them;
//   * LDS images are fragment-shaped: A = weights [octet][tap][m][8 channels], B = activations [octet][pixel slot][8 channels],
//     so a lane's MFMA operand (8 consecutive k of its row / column) is ONE ds_read_b128 and consecutive lanes read consecutive
//     16-byte slots (conflict-free);
//   * the f32 NCHW -> [pixel][8 channels] transposition happens in registers: a staging thread owns one channel OCTET of one
//     pixel quad (1x1: eight 16-byte loads) or of one halo element (3x3: eight dword loads) and writes 16-byte slots; the slots
//     of a 1x1 tile are XOR-swizzled so that the four slots a thread writes do not collide with its neighbours' banks.
#include <algorithm>

#include "common.h"
#include "igemm.h"

namespace s2k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 1x1 tiles: physical 16-byte slot of logical pixel slot s (a staging thread writes slots 4j .. 4j+3: without the swizzle the
// eight lanes of a ds_write_b128 group hit two 16-byte columns of the 128-byte bank window, 4-way)
__device__ __forceinline__ int swz(int s) { return s ^ ((s >> 3) & 3); }

// PRO: S2K_PRO_NONE / RELU / SILU / AFFINE (1x1), NONE / RELU (3x3).  GATE: SE gate [B][C1] multiplies the activated value (1x1).
// SCATTER (1x1): S2K_MODE_CONVT_SCATTER - rows m = (co, dy, dx) are stored to Y[b][co][2y + dy][2x + dx] (ConvTranspose2d k2 s2).
// X16 (no prologue, one source): X1 is stored as bf16 [B][C1][HW] (a BN_BWD_APPLY with OUT_BF16 wrote it): 1x1 - four pixels of a channel
// are one 8-byte load; 3x3 - a halo element is one 16-bit load; the operand units are assembled from the halves - no conversion,
// the values are the ones the f32 path would round to.
template <int BMODE, int WVM, int WM, int WN, int KCH, int R, int XW, int PRO, bool GATE, bool SCATTER = false, bool X16 = false>
__global__ void __launch_bounds__(256, 2) conv_bf16_kernel(const ConvP p) {
    constexpr bool PIX = BMODE == BM_PIX;
    static_assert(!X16 || (PRO == S2K_PRO_NONE && !GATE && !SCATTER), "bf16 X1: no prologue");
    constexpr int NT = 256;
    constexpr int WVN = 4 / WVM;
    constexpr int BM = WVM * WM * 32, BN = WVN * WN * 32;
    constexpr int TT = PIX ? 1 : 9;
    constexpr int NO = KCH / 8;                              // channel octets per chunk
    constexpr int WS = XW + 2, IR = R + 2;                   // 3x3 stride-1 halo tile
    constexpr int USED = PIX ? BN : IR * WS;                 // B slots per octet
    constexpr int A_UNITS = NO * TT * BM;                    // 16-byte units
    constexpr int B_UNITS = NO * USED;
    constexpr int NA = (A_UNITS + NT - 1) / NT;
    constexpr int B_ITEMS = PIX ? (BN / 4) * NO : B_UNITS;   // staging items: (pixel quad | halo element) x octet
    constexpr int NBI = (B_ITEMS + NT - 1) / NT;
    static_assert(KCH % 16 == 0, "a k-step is 16 channels");
    static_assert(PIX || R * XW <= BN, "3x3 tile");
    static_assert(PIX || (!GATE && !SCATTER && (PRO == S2K_PRO_NONE || PRO == S2K_PRO_RELU)), "3x3: BatchNorm + ReLU prologue at most");
    static_assert(KCH <= 64 || PIX, "128-channel chunks: 1x1 only");
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_b[];
    u32x4* As = smem_b;
    u32x4* Bs = smem_b + A_UNITS;
    float* tab = reinterpret_cast<float*>(Bs + B_UNITS);     // [2][Ctp] scale / shift of every (concat) channel, zero beyond Ctot

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int m0 = mt * BM;
    const int HWo = p.HO * p.WO;
    const int nchunks = (p.Ctot + KCH - 1) / KCH;
    const int Ctp = nchunks * KCH;
    // split-K (deep 1x1 convs over few pixels): this workgroup reduces chunks [ch_begin, ch_end) and leaves a partial tile
    const int ch_begin = blockIdx.y * p.chunks_per_split;
    const int ch_end = min(nchunks, ch_begin + p.chunks_per_split);
    const int tx = PIX ? 0 : nt % p.tiles_x;
    const int ty = PIX ? 0 : (nt / p.tiles_x) % p.tiles_y;
    const int sb = PIX ? 0 : nt / (p.tiles_x * p.tiles_y);
    const int y0 = ty * R, x0 = tx * XW;
    const int MP = p.w_st;                                   // packed row count (M rounded up to 128)
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.wtb);
    const int kp8 = ((p.Ctot + 63) / 64) * 8;                // octets of the packed K (WEIGHT_PACK pads K to a multiple of 64)

    // ---- staging geometry: fixed for the whole K loop ---------------------------------------------------------------------
    uint32_t bvoff[NBI];        // byte offset of the item's pixel quad / halo element in a channel plane (image-relative), or OOB
    float bound[PIX ? 1 : NBI]; // 3x3: +inf inside the image, 0 in the zero padding
    int b_dst[NBI];             // LDS unit of the item (pixel quad: first of its 4 slots, unswizzled)
    int b_co[NBI];              // octet of the item
    int gate_b[PIX ? NBI : 1];  // 1x1: image of the pixel quad (SE gate row)
    int img_b = 0;
    if (PIX) {
        img_b = (nt * BN) / p.HW;             // descriptors based at the tile's first image (see igemm.hip)
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int it = tid + NT * i;
            const int j4 = it % (BN / 4), co = it / (BN / 4);
            const int n = nt * BN + 4 * j4;
            const bool ok = it < B_ITEMS && n < p.Ntot;
            const int nn = ok ? n : 0;
            const int b = nn / p.HW, pp = nn - b * p.HW;
            bvoff[i] = ok ? (uint32_t)((int64_t)(b - img_b) * p.C1 * p.HW + pp) * (X16 ? 2u : 4u) : BUF_OOB;
            b_dst[i] = co * BN + 4 * j4;
            b_co[i] = co;
            gate_b[PIX ? i : 0] = b;
        }
    } else {
        img_b = sb;
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int it = tid + NT * i;
            const int e = it % USED, co = it / USED;
            const int iy = y0 - p.PT + e / WS, ix = x0 - p.PL + e % WS;
            const bool ok = it < B_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            bvoff[i] = ok ? (uint32_t)(iy * p.W + ix) * (X16 ? 2u : 4u) : BUF_OOB;
            bound[PIX ? 0 : i] = ok ? __builtin_inff() : 0.0f;
            b_dst[i] = co * USED + e;
            b_co[i] = co;
        }
    }
    const int64_t x1_img = (int64_t)p.C1 * p.HW, x2_img = (int64_t)p.C2 * p.HW;
    const rsrc_t rx1 = X16 ? make_rsrc(reinterpret_cast<const uint16_t*>(p.x1) + img_b * x1_img, (p.B - img_b) * x1_img * 2)
                           : make_rsrc(p.x1 + img_b * x1_img, (p.B - img_b) * x1_img * 4);
    const rsrc_t rx2 = make_rsrc(p.x2 ? p.x2 + img_b * x2_img : p.x1, p.x2 ? (p.B - img_b) * x2_img * 4 : 0);
    const uint32_t cs4 = (uint32_t)p.HW * (X16 ? 2u : 4u);      // bytes between channel planes

    // ---- prologue table -----------------------------------------------------------------------------------------------------
    if (PRO != S2K_PRO_NONE) {
        for (int c = tid; c < Ctp; c += NT) {
            float sc = 0.0f, sh = 0.0f;
            if (c < p.C1) { sc = p.bnv1[c]; sh = p.bnv1[p.C1 + c]; }
            else if (c < p.Ctot) { sc = p.bnv2[c - p.C1]; sh = p.bnv2[p.C2 + c - p.C1]; }
            tab[c] = sc;
            tab[Ctp + c] = sh;
        }
    }

    u32x4 areg[NA];
    f32x4 bq[(PIX && !X16) ? NBI : 1][8];          // 1x1: eight channels x four pixels per item
    float2 bh[X16 ? NBI : 1][8];                   // X16: the same as four bf16 (two dwords) per channel
    float bs[PIX ? 1 : NBI][8];          // 3x3: eight channels of one halo element per item
    float greg[(PIX && GATE) ? NBI : 1][8];

    auto fetch = [&](int c0) {
        // A: units u = (o * TT + tap) * BM + row, contiguous rows of the packed bf16 weights [octet][tap][MP][8]
        const int ob0 = c0 >> 3;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int u = tid + NT * i;
            const int row = u % BM, ot = u / BM;             // ot = o * TT + tap
            if constexpr (KCH > 64) {
                // the packed K is a multiple of 64: the upper half of the last 128-channel chunk may lie past the entry - re-read its
                // last octet (the matching B units are zeroed at the commit)
                const int o = min(ob0 + ot / TT, kp8 - 1), tap = ot % TT;
                if (A_UNITS % NT == 0 || u < A_UNITS) areg[i] = wsrc[((int64_t)o * TT + tap) * MP + m0 + row];
            } else {
                if (A_UNITS % NT == 0 || u < A_UNITS) areg[i] = wsrc[((int64_t)ob0 * TT + ot) * MP + m0 + row];
            }
        }
        if (PIX) {
#pragma unroll
            for (int i = 0; i < NBI; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = min(c0 + 8 * b_co[i] + q, p.Ctot - 1);   // channels past Ctot meet zero rows of the packed weights
                    if constexpr (X16) bh[X16 ? i : 0][q] = bload2(rx1, bvoff[i] + (uint32_t)c * cs4);
                    else bq[(PIX && !X16) ? i : 0][q] = bload4(rx1, bvoff[i] + (uint32_t)c * cs4);
                    if (GATE) greg[(PIX && GATE) ? i : 0][q] = p.gate1[gate_b[PIX ? i : 0] * p.C1 + c];
                }
        } else {
            const bool first = c0 < p.C1;                    // a chunk never straddles the two sources (host: C1 % KCH == 0)
            const int cb = first ? c0 : c0 - p.C1, cn = first ? p.C1 : p.C2;
#pragma unroll
            for (int i = 0; i < NBI; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = min(cb + 8 * b_co[i] + q, cn - 1);
                    if constexpr (X16) bs[PIX ? 0 : i][q] = __builtin_bit_cast(float, bload_u16(rx1, bvoff[i] + (uint32_t)c * cs4));   // (C2 = 0)
                    else bs[PIX ? 0 : i][q] = first ? bload(rx1, bvoff[i] + (uint32_t)c * cs4) : bload(rx2, bvoff[i] + (uint32_t)c * cs4);
                }
        }
    };
    auto commit = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int u = tid + NT * i;
            if (A_UNITS % NT == 0 || u < A_UNITS) As[u] = areg[i];
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            if (B_ITEMS % NT != 0 && tid + NT * i >= B_ITEMS) continue;
            float sc[8], sh[8];
            if (PRO != S2K_PRO_NONE) {
                const f32x4* ts = reinterpret_cast<const f32x4*>(tab + c0 + 8 * b_co[i]);
                const f32x4* th = reinterpret_cast<const f32x4*>(tab + Ctp + c0 + 8 * b_co[i]);
                const f32x4 s0 = ts[0], s1 = ts[1], h0 = th[0], h1 = th[1];
#pragma unroll
                for (int q = 0; q < 4; ++q) { sc[q] = s0[q]; sc[4 + q] = s1[q]; sh[q] = h0[q]; sh[4 + q] = h1[q]; }
            }
            if constexpr (X16 && PIX) {
                // dword d of channel q holds pixels 2d (low half) and 2d + 1 (high half): unit e = pixel e of the eight channels
                uint32_t lo[8], hi[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { lo[q] = __builtin_bit_cast(uint32_t, bh[X16 ? i : 0][q].x); hi[q] = __builtin_bit_cast(uint32_t, bh[X16 ? i : 0][q].y); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {       // even channel in the low half, odd channel in the high half (the compiler picks v_perm_b32 / v_and_or)
                        const uint32_t a = (e < 2) ? lo[2 * k] : hi[2 * k], b = (e < 2) ? lo[2 * k + 1] : hi[2 * k + 1];
                        w[k] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                    }
                    if constexpr (KCH > 64) { if (c0 + 8 * b_co[i] >= 8 * kp8) w = u32x4{0u, 0u, 0u, 0u}; }
                    const int s = (b_dst[i] % BN) + e;
                    Bs[(b_dst[i] - (b_dst[i] % BN)) + swz(s)] = w;
                }
            } else if (PIX) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float x = bq[(PIX && !X16) ? i : 0][q][e];
                        if (PRO != S2K_PRO_NONE) x = apply_pro_c<PRO>(x, sc[q], sh[q]);
                        if (GATE) x *= greg[(PIX && GATE) ? i : 0][q];
                        v[q] = x;
                    }
                    u32x4 w = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                    if constexpr (KCH > 64) { if (c0 + 8 * b_co[i] >= 8 * kp8) w = u32x4{0u, 0u, 0u, 0u}; }    // octets past the packed K
                    const int s = (b_dst[i] % BN) + e;
                    Bs[(b_dst[i] - (b_dst[i] % BN)) + swz(s)] = w;
                }
            } else if constexpr (X16) {
                uint32_t hb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) hb[q] = __builtin_bit_cast(uint32_t, bs[PIX ? 0 : i][q]);      // zero-extended halves; padding = 0
                Bs[b_dst[i]] = u32x4{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
            } else {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float x = bs[PIX ? 0 : i][q];            // padding slots were loaded as 0
                    if (PRO == S2K_PRO_RELU) x = __builtin_amdgcn_fmed3f(fmaf(x, sc[q], sh[q]), 0.0f, bound[PIX ? 0 : i]);   // the reference pads ACTIVATED maps
                    v[q] = x;
                }
                u32x4 w = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                Bs[b_dst[i]] = w;
            }
        }
    };

    // ---- consumer geometry ------------------------------------------------------------------------------------------------------
    const int wm0 = (wave / WVN) * (WM * 32), wn0 = (wave % WVN) * (WN * 32);
    int bslot[WN];
#pragma unroll
    for (int rn = 0; rn < WN; ++rn) {
        const int j = wn0 + rn * 32 + l31;
        if (PIX) bslot[rn] = swz(j);
        else bslot[rn] = (j < R * XW) ? (j / XW) * WS + (j % XW) : 0;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    fetch(ch_begin * KCH);
    __syncthreads();                                             // prologue table complete
    commit(ch_begin * KCH);
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        if (ch + 1 < ch_end) fetch((ch + 1) * KCH);              // in flight during the MFMAs below
        // k-step (tap, s): 16 channels = octets 2 s and 2 s + 1; lane half lh takes octet 2 s + lh
#pragma unroll
        for (int tap = 0; tap < ((p.exp & 2) ? 0 : TT); ++tap) {
            const int toff = PIX ? 0 : (tap / 3) * WS + (tap % 3);
#pragma unroll
            for (int s = 0; s < NO / 2; ++s) {
                bf16x8 a[WM], b[WN];
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
                    a[rm] = __builtin_bit_cast(bf16x8, As[((2 * s) * TT + tap) * BM + lh * (TT * BM) + wm0 + rm * 32 + l31]);
#pragma unroll
                for (int rn = 0; rn < WN; ++rn)
                    b[rn] = __builtin_bit_cast(bf16x8, Bs[(2 * s) * USED + lh * USED + bslot[rn] + toff]);
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm], b[rn], acc[rm][rn], 0, 0, 0);
            }
        }
        __syncthreads();                                         // every wave is done reading the image
        if (ch + 1 < ch_end) {
            commit((ch + 1) * KCH);
            __syncthreads();
        }
    }

    // ---- epilogue: f32, as igemm_pc.hip ----------------------------------------------------------------------------------------
    if (p.exp & 1) return;      // tuning builds only (S2K_CV_EXP)
    bool cval[WN];
    int64_t ycol[WN];
#pragma unroll
    for (int rn = 0; rn < WN; ++rn) {
        const int j = wn0 + rn * 32 + l31;
        if (PIX) {
            const int n = nt * BN + j;
            cval[rn] = n < p.Ntot;
            const int nn = cval[rn] ? n : 0;
            const int b = nn / p.HW, pp = nn - b * p.HW;
            if (SCATTER) {
                const int yy = pp / p.W, xq = pp - yy * p.W;
                ycol[rn] = (int64_t)b * p.YC * 4 * p.HW + (int64_t)(2 * yy) * (2 * p.W) + 2 * xq;
            } else {
                ycol[rn] = (int64_t)b * p.YC * HWo + pp;
            }
        } else {
            const int r = j / XW, xx = j % XW;
            cval[rn] = (j < R * XW) && (y0 + r < p.HO) && (x0 + xx < p.WO);
            ycol[rn] = (int64_t)sb * p.YC * HWo + (int64_t)(y0 + r) * p.WO + (x0 + xx);
        }
    }
    if (p.splits > 1) {   // partial tile -> scratch; bias / residual / accumulate / statistics are applied by the split-K tail kernel
        float* part = p.scratch + (int64_t)blockIdx.y * p.y_elems;
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn)
                    if (gm < p.M && cval[rn]) part[ycol[rn] + (int64_t)gm * HWo] = acc[rm][rn][reg];
            }
        return;
    }
    if (SCATTER) {
        // registers 4q .. 4q+3 of a lane are the 2 x 2 output patch (dy, dx) of one output channel co = row / 4
        const int64_t plane = 4 * (int64_t)p.HW;
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gm = m0 + wm0 + rm * 32 + 8 * q + 4 * lh;
                if (gm < p.M) {
                    const int co = gm >> 2;
                    const float bsv = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        if (cval[rn]) {
                            float* dst = p.y + ycol[rn] + (int64_t)co * plane;
                            *reinterpret_cast<float2*>(dst) = make_float2(acc[rm][rn][4 * q + 0] + bsv, acc[rm][rn][4 * q + 1] + bsv);
                            *reinterpret_cast<float2*>(dst + 2 * p.W) = make_float2(acc[rm][rn][4 * q + 2] + bsv, acc[rm][rn][4 * q + 3] + bsv);
                        }
                }
            }
        return;
    }
    if constexpr (!SCATTER && (PIX || XW % 4 == 0)) {
        // ---- transposed store -------------------------------------------------------------------------------------------------------
        // An accumulator register holds ONE pixel of a row per lane: stored as it is, every instruction writes 2 rows x 128 bytes,
        // and the write-heavy layers (40 -> 240-channel expand convs: 126 MB out for 21 MB in) streamed their output at 2.9-3.4 TB/s
        // where a fill reaches 6.8 (tools/exp_cv16.sh: 37 of the stage's 46 us were the stores).  So the tile goes through LDS, 32 rows
        // per wave row at a time, and comes back pixel-major: a lane stores 4 consecutive pixels (16 bytes), an instruction 2 rows x
        // 512 bytes.  Bias / residual / accumulate and the BatchNorm statistics move to the reading side (a row's sums: one DPP
        // reduction over the lanes that hold it, one f64 atomic pair per row and workgroup, fixed order throughout).
        constexpr int BNP = BN + 8;                              // row stride: lane halves (rows r, r + 4) land in different banks
        constexpr int PASS_ROWS = WVM * 32;
        constexpr int G = BN / 4;                                // pixel quads per row
        constexpr int RPI = NT / G;                              // rows per pass of the 256 threads
        static_assert(NT % G == 0 && PASS_ROWS % RPI == 0 && (G == 16 || G == 32 || G == 64), "transposed epilogue geometry");
        float* ct = reinterpret_cast<float*>(smem_b);            // [PASS_ROWS][BNP]  (the image is dead: barrier above)
        const int g4 = tid % G, r0 = tid / G;
        bool gok;
        int64_t gcol;
        {
            const int j = 4 * g4;
            if (PIX) {
                const int n = nt * BN + j;
                gok = n < p.Ntot;
                const int nn = gok ? n : 0;
                const int b = nn / p.HW, pp = nn - b * p.HW;
                gcol = (int64_t)b * p.YC * HWo + pp;
            } else {
                const int r = j / XW, xx = j % XW;
                gok = (j < R * XW) && (y0 + r < p.HO) && (x0 + xx < p.WO);
                gcol = (int64_t)sb * p.YC * HWo + (int64_t)(y0 + r) * p.WO + (x0 + xx);
            }
        }
        double* st = p.stats ? p.stats + (int64_t)(tile % p.nrep) * 2 * p.M : nullptr;
#pragma unroll
        for (int rm = 0; rm < WM; ++rm) {
            if (rm) __syncthreads();                             // the previous pass has been read
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = (wave / WVN) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) ct[rl * BNP + wn0 + rn * 32 + l31] = acc[rm][rn][reg];
            }
            __syncthreads();
#pragma unroll
            for (int r = r0; r < PASS_ROWS; r += RPI) {
                const int gm = m0 + (r >> 5) * (WM * 32) + rm * 32 + (r & 31);
                const bool ok = gok && gm < p.M;
                f32x4 v = *reinterpret_cast<const f32x4*>(ct + r * BNP + 4 * g4);
                float s = 0.0f, q = 0.0f;
                if (ok) {
                    if (p.bias) { const float bsv = p.bias[gm]; v[0] += bsv; v[1] += bsv; v[2] += bsv; v[3] += bsv; }
                    float* dst = p.y + gcol + (int64_t)gm * HWo;
                    if (p.res) { const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + gcol + (int64_t)gm * HWo); v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3]; }
                    if (p.beta) { const f32x4 ov = *reinterpret_cast<const f32x4*>(dst); v[0] += ov[0]; v[1] += ov[1]; v[2] += ov[2]; v[3] += ov[3]; }
                    *reinterpret_cast<f32x4*>(dst) = v;
                    s = (v[0] + v[1]) + (v[2] + v[3]);
                    q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
                }
                if (st) {
                    bool writer;
                    if (G == 16) { s = row16_sum(s); q = row16_sum(q); writer = (lane & 15) == 15; }
                    else if (G == 32) { s = half_sum_hi(s); q = half_sum_hi(q); writer = l31 == 31; }
                    else { s = wave_sum_hi(s); q = wave_sum_hi(q); writer = lane == 63; }
                    if (writer && gm < p.M) {
                        atomic_add_d(st + gm, (double)s);
                        atomic_add_d(st + p.M + gm, (double)q);
                    }
                }
            }
        }
        return;
    }
    float* srow = reinterpret_cast<float*>(smem_b);              // [WVN wave columns][2][BM]  (the image is dead: barrier above)
    const int wn_idx = wave % WVN;
#pragma unroll
    for (int rm = 0; rm < WM; ++rm)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const int gm = m0 + row;
            const bool rok = gm < p.M;
            const float bsv = (p.bias && rok) ? p.bias[gm] : 0.0f;
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) {
                float v = acc[rm][rn][reg] + bsv;
                if (rok && cval[rn]) {
                    float* dst = p.y + ycol[rn] + (int64_t)gm * HWo;
                    if (p.res) v += p.res[ycol[rn] + (int64_t)gm * HWo];
                    if (p.beta) v += *dst;
                    *dst = v;
                    s += v;
                    q += v * v;
                }
            }
            if (p.stats) {
                s = half_sum_hi(s);
                q = half_sum_hi(q);
                if (l31 == 31) {   // each (wave column, row) slot has exactly one writer
                    srow[(wn_idx * 2 + 0) * BM + row] = rok ? s : 0.0f;
                    srow[(wn_idx * 2 + 1) * BM + row] = rok ? q : 0.0f;
                }
            }
        }
    if (p.stats) {
        __syncthreads();
        // partial sums of the wave columns added in a FIXED order (BatchNorm statistics stay reproducible), then one f64 atomic
        // pair per row per workgroup into the statistics replica of this tile
        double* st = p.stats + (int64_t)(tile % p.nrep) * 2 * p.M;
        for (int i = tid; i < 2 * BM; i += NT) {
            const int row = i % BM, which = i / BM, gm = m0 + row;
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < WVN; ++w) tot += srow[(w * 2 + which) * BM + row];
            if (gm < p.M) atomic_add_d(st + which * p.M + gm, (double)tot);
        }
    }
}

// -------------------------------------------------------------------------------------------------
template <int BMODE, int WVM, int WM, int WN, int KCH, int R, int XW, int PRO, bool GATE, bool SCATTER = false, bool X16 = false>
static int launch_b16(ConvP& p, int n_ntiles, hipStream_t st) {
    constexpr bool PIX = BMODE == BM_PIX;
    constexpr int WVN = 4 / WVM;
    constexpr int BM = WVM * WM * 32, BN = WVN * WN * 32;
    constexpr int TT = PIX ? 1 : 9;
    constexpr int NO = KCH / 8;
    constexpr int USED = PIX ? BN : (R + 2) * (XW + 2);
    constexpr size_t img = (size_t)(NO * TT * BM + NO * USED) * 16;
    static_assert(img <= 160 * 1024, "LDS image");
    static_assert((size_t)WVN * 2 * BM * sizeof(float) <= img, "statistics rows fit in the image");
    p.n_mtiles = cdiv(p.M, BM);
    const int nchunks = cdiv(p.Ctot, KCH);
    constexpr size_t ct_bytes = (!SCATTER && (PIX || XW % 4 == 0)) ? (size_t)WVM * 32 * (BN + 8) * sizeof(float) : 0;   // transposed store
    const size_t lds = std::max(img + (PRO != S2K_PRO_NONE ? (size_t)2 * nchunks * KCH * sizeof(float) : 0), ct_bytes);
    if (lds > 160 * 1024) return 1;
    {   // 32-bit buffer offsets: image-local tiles need one image < 2 GiB, tiles that may straddle images the whole tensor
        // (descriptors are based at the first image a tile touches)
        const int64_t span = (!PIX || (p.HW % BN) == 0) ? 1 : std::min<int64_t>(p.B, (BN - 2) / p.HW + 2);
        const int64_t img1 = (int64_t)p.C1 * p.H * p.W * 4, img2 = (int64_t)p.C2 * p.H * p.W * 4;
        const int64_t need = std::max(img1, img2) * span;
        if (need >= 0x7ffffff0ll) { set_error("conv: the %lld image(s) one tile touches exceed 2 GiB (%lld B)", (long long)span, (long long)need); return S2K_EINVAL; }
    }
    const int64_t blocks = (int64_t)p.n_mtiles * n_ntiles;
    if (blocks <= 0 || blocks > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)blocks); return S2K_EINVAL; }
    p.n_tiles = (int)blocks;
    p.splits = 1;
    p.chunks_per_split = nchunks;
    p.y_elems = (int64_t)p.B * p.YC * p.HO * p.WO;
    static const int split_max_blocks = tune_int("S2K_B16_SPLIT_MAX_BLOCKS", 512);
    if (PIX && !SCATTER && p.scratch && blocks <= split_max_blocks && nchunks >= 8) {
        // Few tiles and a long reduction (the 1x1 convs of the 8 x 8 / 16 x 16 blocks: 160 workgroups each walking 29 - 48
        // chunks one memory round trip at a time): cut K so that ~1,000 workgroups share the round trips; the partial tiles
        // are added in a fixed order by the split-K tail (igemm.hip), which also applies bias / statistics.  SCRATCH holds 8.
        int splits = (int)std::min<int64_t>(8, cdiv64(1024, blocks));
        splits = std::min(splits, nchunks / 4);
        if (splits > 1) {
            p.chunks_per_split = cdiv(nchunks, splits);
            p.splits = cdiv(nchunks, p.chunks_per_split);
        }
    }
    auto kern = conv_bf16_kernel<BMODE, WVM, WM, WN, KCH, R, XW, PRO, GATE, SCATTER, X16>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.splits), dim3(256), lds, st, p);
    if (p.splits > 1) launch_splitk_reduce(p, st);
    g_s2k_variant = 2;
    return S2K_OK;
}

// tile height by M: 64 rows when 128 would pad M by more than 12 % (32-row tiles, 1 x 4 waves over 256 pixels, are instantiated
// by the callers: their pixel tile differs)
template <int BMODE, int KCH, int R, int XW, int PRO, bool GATE, bool SCATTER = false, bool X16 = false>
static int launch_b16_bm(ConvP& p, int n128, hipStream_t st) {
    if (p.M <= 32) return 1;
    const bool big = p.M > 64 && (double)cdiv(p.M, 128) * 128 / p.M <= 1.12;
    if (big) return launch_b16<BMODE, 2, 2, 2, KCH, R, XW, PRO, GATE, SCATTER, X16>(p, n128, st);
    return launch_b16<BMODE, 2, 1, 2, KCH, R, XW, PRO, GATE, SCATTER, X16>(p, n128, st);
}

template <int KCH, int PRO, bool GATE, bool X16 = false>
static int launch_b16_pix(ConvP& p, hipStream_t st) {
    if constexpr (KCH <= 64) {
        if (p.M <= 32) return launch_b16<BM_PIX, 1, 1, 2, KCH, 1, 64, PRO, GATE, false, X16>(p, cdiv(p.Ntot, 256), st);
    }
    // few pixels (the 8 x 8 / 16 x 16 maps of the deep blocks): 128-pixel tiles would leave most CUs without a workgroup and each
    // of the few with a long serial K loop; 64 x 64 tiles give 4x the workgroups
    const int n128 = cdiv(p.Ntot, 128);
    if ((int64_t)n128 * cdiv(p.M, 128) < 200) return launch_b16<BM_PIX, 2, 1, 1, KCH, 1, 64, PRO, GATE, false, X16>(p, cdiv(p.Ntot, 64), st);
    return launch_b16_bm<BM_PIX, KCH, 1, 64, PRO, GATE, false, X16>(p, n128, st);
}

// S2K_OK = launched, 1 = not one of its shapes (the caller takes the f32 kernels), < 0 = error
int launch_conv_bf16(ConvP& p, hipStream_t st) {
    if (!p.wtb || p.S != 1 || p.HO != p.H || p.WO != p.W) return 1;
    const int T = p.KH * p.KW;
    if (p.x1_bf16 && p.mode != S2K_MODE_CONV) return 1;
    if (p.mode == S2K_MODE_CONVT_SCATTER) {
        // ConvTranspose2d(k2, s2) forward: a 1x1 contraction with rows (co, dy, dx) and a scattering epilogue
        if (T != 1 || p.C2 != 0 || (p.HW & 3) || p.gate1 || p.M <= 32 || (p.M & 3)) return 1;
        const int n128 = cdiv(p.Ntot, 128);
        if (p.pro1 == S2K_PRO_RELU) return launch_b16_bm<BM_PIX, 64, 1, 64, S2K_PRO_RELU, false, true>(p, n128, st);
        if (p.pro1 == S2K_PRO_SILU) return launch_b16_bm<BM_PIX, 64, 1, 64, S2K_PRO_SILU, false, true>(p, n128, st);
        return 1;
    }
    if (p.mode != S2K_MODE_CONV) return 1;
    if (T == 1) {
        if (p.C2 != 0 || (p.HW & 3) || p.PT || p.PL) return 1;
        // deep reductions (the project / data-gradient convs of the 8 x 8 - 32 x 32 blocks, ViT Linears): with bf16 MFMAs a chunk's
        // multiply is short and a workgroup's K loop is a chain of load round trips - 128-channel chunks halve it
        static const int deep_min = tune_int("S2K_B16_DEEP_MIN", 512);
        // (measured: 8 x 8 / 16 x 16 maps 43 -> 22 us, 34 -> 18, 20 -> 10; large pixel counts are matrix-core / bandwidth-bound and lose
        // 4 - 8 % to the larger LDS image, so only problems of at most 1,024 128 x 128 tiles take it; the 32-row tile would spill)
        const bool deep = p.Ctot >= deep_min && p.M > 32 && (int64_t)cdiv(p.Ntot, 128) * cdiv(p.M, 128) <= 1024;
        if (p.gate1) {
            if (p.pro1 == S2K_PRO_SILU) return deep ? launch_b16_pix<128, S2K_PRO_SILU, true>(p, st) : launch_b16_pix<64, S2K_PRO_SILU, true>(p, st);
            return 1;
        }
        if (p.x1_bf16) {      // X1 stored as bf16 (opdefs CONV.X1_BF16): planned only for this shape class; nothing else reads such a tensor
            if (p.gate1 || p.pro1 != S2K_PRO_NONE || (p.HW & 7)) { set_error("conv: X1_BF16 is for plain 1x1 stages (no prologue, H*W % 8 == 0)"); return S2K_EINVAL; }
            return deep ? launch_b16_pix<128, S2K_PRO_NONE, false, true>(p, st) : launch_b16_pix<64, S2K_PRO_NONE, false, true>(p, st);
        }
        switch (p.pro1) {
            case S2K_PRO_NONE: return deep ? launch_b16_pix<128, S2K_PRO_NONE, false>(p, st) : launch_b16_pix<64, S2K_PRO_NONE, false>(p, st);
            case S2K_PRO_RELU: return launch_b16_pix<64, S2K_PRO_RELU, false>(p, st);
            case S2K_PRO_SILU: return launch_b16_pix<64, S2K_PRO_SILU, false>(p, st);
            case S2K_PRO_AFFINE: return launch_b16_pix<64, S2K_PRO_AFFINE, false>(p, st);
            default: return 1;
        }
    }
    if (T != 9 || p.KH != 3 || p.PT != 1 || p.PL != 1 || p.gate1) return 1;
    if (p.pro1 != S2K_PRO_NONE && p.pro1 != S2K_PRO_RELU) return 1;
    if (p.x1_bf16 && (p.pro1 != S2K_PRO_NONE || p.C2 != 0)) return 1;      // (the f32 launcher reports the error)
    if (p.C2 > 0 && (p.pro2 != p.pro1 || (p.C1 % 16) != 0)) return 1;
    auto tiles = [&](int r, int xw) {
        p.R = r; p.XW = xw; p.IR = r + 2; p.IC = xw + 2; p.WS = xw + 2; p.CS = p.IR * p.WS;
        p.tiles_x = cdiv(p.WO, xw);
        p.tiles_y = cdiv(p.HO, r);
        return p.B * p.tiles_x * p.tiles_y;
    };
    const bool relu = p.pro1 == S2K_PRO_RELU;
#define B16_3X3(RR, XX) { const int n = tiles(RR, XX); \
        if (p.x1_bf16) return launch_b16_bm<BM_SPATIAL, 16, RR, XX, S2K_PRO_NONE, false, false, true>(p, n, st); \
        return relu ? launch_b16_bm<BM_SPATIAL, 16, RR, XX, S2K_PRO_RELU, false>(p, n, st) \
                    : launch_b16_bm<BM_SPATIAL, 16, RR, XX, S2K_PRO_NONE, false>(p, n, st); }
    if (p.M <= 32) {
        if (p.WO < 64 || p.WO % 64 != 0) return 1;
        const int n = tiles(4, 64);
        if (p.x1_bf16) return launch_b16<BM_SPATIAL, 1, 1, 2, 16, 4, 64, S2K_PRO_NONE, false, false, true>(p, n, st);
        return relu ? launch_b16<BM_SPATIAL, 1, 1, 2, 16, 4, 64, S2K_PRO_RELU, false>(p, n, st)
                    : launch_b16<BM_SPATIAL, 1, 1, 2, 16, 4, 64, S2K_PRO_NONE, false>(p, n, st);
    }
    if (p.WO >= 64 && p.WO % 64 == 0) B16_3X3(2, 64)
    if (p.WO == 32) B16_3X3(4, 32)
    if (p.WO == 16) B16_3X3(8, 16)
    if (p.WO == 56 || p.WO == 112 || p.WO == 224) B16_3X3(2, 56)
    if (p.WO == 28) B16_3X3(4, 28)
    if (p.WO == 14) B16_3X3(8, 14)
#undef B16_3X3
    return 1;
}

}  // namespace s2k
