// Implicit-GEMM convolution on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the dense convs of the hot path (reference efficientnet_unet.py):
//   Conv2dSamePadding 1x1 / 3x3-s2 stem (:288-297), nn.Conv2d 3x3 pad 1 (_double_conv :168-176),
//   nn.ConvTranspose2d k2 s2 (:114,:120), and their data gradients (ATen convolution_backward),
// with BatchNorm-apply + SiLU/ReLU + SE gate folded into the operand *load* (prologue), channel
// concat as two source pointers, bias + BatchNorm batch statistics folded into the epilogue.
//
// GEMM view (NCHW, exact f32):  D[m][n] = sum_k A[m][k] * Bop[k][n]
//   m = output channel (weights, A), n = output pixel (lanes -> coalesced 128-B row stores),
//   k = (input channel, tap).  MFMA 32x32x2: lane l holds A[m = l&31][k = l>>5] and
//   Bop[k = l>>5][n = l&31]; D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].
// LDS: As[k][m] (row stride BM+1) and Bs[k][pixels]; with pixels on the lanes every ds_read_b32
// is 32 consecutive words per half-wave (conflict-free), and the im2col of a KxK conv is just a
// per-tap constant added to the per-lane pixel offset inside a halo tile.
//
// Pipeline: the operands of K-chunk i+1 are fetched global->registers *before* the MFMAs of chunk
// i are issued and written to LDS after them (one LDS buffer, two barriers per chunk), so HBM/L2
// latency hides under 64..288 MFMAs per wave; all index arithmetic of the stagers is incremental
// or compile-time (no runtime divisions in the K loop).  Workgroup ids are remapped so that the
// m-tiles sharing one activation tile run on the same XCD (shared L2).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "igemm.h"

namespace s2k {

// BVEC (1x1 stager only): a lane moves 4 consecutive pixels of one channel (one 16-byte load, one ds_write_b128, one
// gate load) instead of one: the per-element address / validity / LDS-store work of the prologue-heavy 1x1 convs drops 4x.
// Needs H*W % 4 == 0 (a group of 4 never straddles two images) and contiguous pixels (not the 2x2 gather mode).
template <int BMODE, int TT, int WM, int WN, int WVM, int WVN, int KCH, int EPT, int MINW, bool BVEC = false>
__global__ void __launch_bounds__(NTHREADS, MINW) conv_igemm_kernel(const ConvP p) {
    constexpr int BM = WM * WVM * 32;
    constexpr int BN = WN * WVN * 32;
    constexpr int AS = BM;                                                   // A rows are copied whole: no padding needed
    constexpr int KT = KCH * TT;
    constexpr int A4 = KT * BM / 4;                                          // float4 pieces of an A tile
    constexpr int NA = (A4 + NTHREADS - 1) / NTHREADS;                       // float4 pieces per thread per chunk
    constexpr int NB = (BMODE == BM_PIX) ? KCH * BN / NTHREADS : KCH * EPT;  // B elements per thread per chunk
    constexpr int A_FLOATS = (KT * AS + 3) & ~3;
    static_assert(WVM * WVN == 4, "4 waves per workgroup");
    static_assert(!BVEC || BMODE == BM_PIX, "vector stager is for the 1x1 path");
    constexpr int LPR = BVEC ? BN / 4 : BN;                   // stager lanes per B row
    constexpr int KSTEPV = (NTHREADS >= LPR) ? NTHREADS / LPR : 1;
    constexpr int NBV = BVEC ? KCH / KSTEPV : 1;              // float4 pieces per thread per chunk
    static_assert(!BVEC || (KCH % KSTEPV == 0 && NTHREADS % LPR == 0), "vector stager shape");
    static_assert(BMODE != BM_PIX || (NTHREADS % BN == 0 || BN % NTHREADS == 0), "pixel tile vs threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + A_FLOATS;
    // BatchNorm scale/shift of every (concat) input channel, staged once: the commit step below must
    // not contain loads behind branches (each would be a dependent L2 round trip per element)
    float* ssc = Bs + p.b_floats;   // [Ctot] scale
    float* ssh = ssc + p.Ctot;      // [Ctot] shift
    const bool has_pro = true;  // ssc/ssh are always staged (scale 1, shift 0 where a source has no prologue)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WVN) * (WM * 32);
    const int wn0 = (wave % WVN) * (WN * 32);
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int ksplit = blockIdx.y;   // split-K: this workgroup reduces chunks [ch_begin, ch_end) only
    const int mt = tile % p.n_mtiles;
    const int nt = tile / p.n_mtiles;
    const int m0 = mt * BM;
    const bool gather = (p.mode == S2K_MODE_GATHER2X2);
    const bool scatter = (p.mode == S2K_MODE_CONVT_SCATTER);
    const int HWo = p.HO * p.WO;

    // ---- per-lane output columns (MFMA B operand columns == epilogue pixels) -----------------
    int boff[WN];
    bool cval[WN];
    int64_t ycol[WN];
    // ---- B stager bookkeeping ---------------------------------------------------------------
    int sp_goff[(BMODE == BM_SPATIAL) ? EPT : 1];
    int64_t st_off1 = 0, st_off2 = 0, st_cstride = 0;
    int st_gate = 0;
    bool st_valid = false;
    int img_b = 0;                                                  // image the activation descriptors are based at

    if (BMODE == BM_PIX) {
        const int n0 = nt * BN;
#pragma unroll
        for (int rn = 0; rn < WN; ++rn) {
            const int j = wn0 + rn * 32 + l31;
            const int n = n0 + j;
            boff[rn] = j;
            cval[rn] = n < p.Ntot;
            const int nn = cval[rn] ? n : 0;
            const int b = nn / p.HW, pp = nn - b * p.HW;
            if (scatter) {
                const int yy = pp / p.W, xx = pp - yy * p.W;
                ycol[rn] = (int64_t)b * p.YC * 4 * p.HW + (int64_t)(2 * yy) * (2 * p.W) + 2 * xx;
            } else {
                ycol[rn] = (int64_t)b * p.YC * HWo + pp;
            }
        }
        const int j = BVEC ? 4 * (tid % LPR) : tid % BN;
        const int n = n0 + j;
        st_valid = n < p.Ntot;
        const int nn = st_valid ? n : 0;
        const int b = nn / p.HW, pp = nn - b * p.HW;
        st_gate = b * p.C1;
        // the descriptors below are based at the image of the tile's first pixel, so offsets only span the images one tile
        // touches (one when H*W % BN == 0, else (BN - 2) / HW + 2): tensors of more than 2 GiB — a 16 x 768 x 224 x 224
        // gradient — are addressable with 32-bit offsets
        img_b = n0 / p.HW;
        const int brel = b - img_b;
        if (gather) {  // X1 is [B][C1/4][2H][2W]; pseudo-channel k=(co,dy,dx)
            const int yy = pp / p.W, xx = pp - yy * p.W;
            st_off1 = (int64_t)brel * (p.C1 / 4) * 4 * p.HW + (int64_t)(2 * yy) * (2 * p.W) + 2 * xx;
        } else {
            st_off1 = (int64_t)brel * p.C1 * p.HW + pp;
            st_off2 = (int64_t)brel * p.C2 * p.HW + pp;
        }
        st_cstride = p.HW;
    } else {
        const int tx = nt % p.tiles_x;
        const int ty = (nt / p.tiles_x) % p.tiles_y;
        const int sb = nt / (p.tiles_x * p.tiles_y);
        const int y0 = ty * p.R, x0 = tx * p.XW;
#pragma unroll
        for (int rn = 0; rn < WN; ++rn) {
            const int j = wn0 + rn * 32 + l31;
            const int r = j / p.XW, xx = j - r * p.XW;
            cval[rn] = (r < p.R) && (y0 + r < p.HO) && (x0 + xx < p.WO);
            boff[rn] = cval[rn] ? (r * p.S) * p.WS + xx * p.S : 0;
            ycol[rn] = (int64_t)sb * p.YC * HWo + (int64_t)(y0 + r) * p.WO + (x0 + xx);
        }
        const int iy0 = y0 * p.S - p.PT, ix0 = x0 * p.S - p.PL;
        const int used = p.IR * p.WS;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + NTHREADS * i;
            int g = -1;
            if (e < used) {
                const int rr = e / p.WS, cc = e - rr * p.WS;
                const int iy = iy0 + rr, ix = ix0 + cc;
                if (cc < p.IC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) g = iy * p.W + ix;
            }
            sp_goff[(BMODE == BM_SPATIAL) ? i : 0] = g;
        }
        img_b = sb;              // a 3x3 tile always lies in one image: image-local descriptors
        st_off1 = 0;
        st_off2 = 0;
        st_cstride = (int64_t)p.H * p.W;
        st_gate = sb * p.C1;
    }

    // bounds-checked descriptors of the activation sources (+ the SE gate; zero-sized when absent), based at image
    // img_b and covering the rest of the tensor from there (clamped to 2 GiB: image-local tiles never reach that far)
    const int64_t x1_img = gather ? (int64_t)(p.C1 / 4) * 4 * p.HW : (int64_t)p.C1 * p.H * p.W;   // elements per image
    const int64_t x2_img = (int64_t)p.C2 * p.H * p.W;
    const rsrc_t rx1 = make_rsrc(p.x1 + img_b * x1_img, (p.B - img_b) * x1_img * 4);
    const rsrc_t rx2 = make_rsrc(p.x2 ? p.x2 + img_b * x2_img : p.x1, p.x2 ? (p.B - img_b) * x2_img * 4 : 0);
    const rsrc_t rgt = make_rsrc(p.gate1 ? p.gate1 : p.x1, p.gate1 ? (int64_t)p.B * p.C1 * 4 : 0);
    const uint32_t st_voff1 = (uint32_t)st_off1 * 4u, st_voff2 = (uint32_t)st_off2 * 4u, st_cs4 = (uint32_t)st_cstride * 4u;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // compile-time B row stride (3x3: the halo capacity of the stager, not the halo actually used): k-step and tap offsets of
    // the operand reads fold into ds_read immediates — the f32 MFMA holds the vector issue port, address VALU is not free
    constexpr int CSB = (BMODE == BM_PIX) ? BN : NTHREADS * EPT;
    const int nchunks = (p.Ctot + KCH - 1) / KCH;
    const int used_sp = p.IR * p.WS;

    f32x4 areg[NA];
    float breg[BVEC ? 1 : NB];
    float greg[(BMODE == BM_PIX && !BVEC) ? NB : 1];   // SE gate of the fetched elements (1x1 project conv only)
    f32x4 bvec[BVEC ? NBV : 1];
    float gvec[BVEC ? NBV : 1];
    if (has_pro) {
        for (int c = tid; c < p.Ctot; c += NTHREADS) {
            float sc = 1.0f, sh = 0.0f;
            if (c < p.C1) { if (p.pro1 != S2K_PRO_NONE) { sc = p.bnv1[c]; sh = p.bnv1[p.C1 + c]; } }
            else if (p.pro2 != S2K_PRO_NONE) { sc = p.bnv2[c - p.C1]; sh = p.bnv2[p.C2 + c - p.C1]; }
            ssc[c] = sc;
            ssh[c] = sh;
        }
    }
    // A tile = rows [c0*TT, c0*TT + KT) x columns [m0, m0 + BM) of the packed weights (WEIGHT_PACK: row
    // stride w_st = MP, zero padded in both directions): affine 16-byte loads, no bounds checks
    const float* a_src = p.wt + (int64_t)(tid / (BM / 4)) * p.w_st + m0 + 4 * (tid % (BM / 4));
    constexpr int A_ROWS_PER_PASS = NTHREADS / (BM / 4);

    // ---------------- global -> registers ----------------------------------------------------------
    auto fetch = [&](int c0) {
        const float* src = a_src + (int64_t)c0 * TT * p.w_st;
#pragma unroll
        for (int i = 0; i < NA; ++i)  // the last partial pass may read (never store) rows past the tile: the
            areg[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)i * A_ROWS_PER_PASS * p.w_st);  // WPACK buffer has slack
        // B: raw values (prologue applied at LDS-store time) through bounds-checked buffer loads: invalid
        // elements are an out-of-range OFFSET (hardware returns 0), so every load is unconditional and all
        // of a chunk's loads are in flight together
        if constexpr (BVEC) {
            const int kc0 = tid / LPR;
            const uint32_t v1 = st_valid ? st_voff1 : BUF_OOB, v2 = st_valid ? st_voff2 : BUF_OOB;   // OOB + channel offset stays OOB
            const uint32_t vg = st_valid ? (uint32_t)st_gate * 4u : BUF_OOB;
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const int c = min(c0 + kc0 + i * KSTEPV, p.Ctot - 1);   // channels past Ctot meet zero rows of the packed weights
                const uint32_t off = (c < p.C1) ? v1 + (uint32_t)c * st_cs4 : v2 + (uint32_t)(c - p.C1) * st_cs4;
                bvec[i] = bload4(c < p.C1 ? rx1 : rx2, off);
                gvec[i] = bload(rgt, vg + (uint32_t)min(c, p.C1 - 1) * 4u);
            }
        } else if (BMODE == BM_PIX) {
            constexpr int KSTEP = (NTHREADS >= BN) ? NTHREADS / BN : 1;
            const int kc0 = __builtin_amdgcn_readfirstlane(tid / BN);
            // the pixel part (or "out of range") is the per-lane vector offset, the channel part a SCALAR offset (the channel
            // of a pass is wave-uniform), clamped into the tensor: channels past Ctot meet zero rows of the packed weights
            const uint32_t v1 = st_valid ? st_voff1 : BUF_OOB, v2 = st_valid ? st_voff2 : BUF_OOB;
            const uint32_t vg = st_valid ? (uint32_t)st_gate * 4u : BUF_OOB;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = min(c0 + kc0 + i * KSTEP, p.Ctot - 1);
                if (gather) {
                    breg[i] = bload_s(rx1, v1, (uint32_t)(c >> 2) * 16u * p.HW + ((c >> 1) & 1) * 8u * p.W + (c & 1) * 4u);
                } else if (c < p.C1) {
                    breg[i] = bload_s(rx1, v1, (uint32_t)c * st_cs4);
                } else {
                    breg[i] = bload_s(rx2, v2, (uint32_t)(c - p.C1) * st_cs4);
                }
                greg[(BMODE == BM_PIX) ? i : 0] = bload_s(rgt, vg, (uint32_t)min(c, p.C1 - 1) * 4u);   // zero-sized descriptor without a gate
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                // halo slot (or "outside the image") = per-lane vector offset; image + channel = SCALAR offset, the channel
                // clamped into the tensor (channels past Ctot meet zero rows of the packed weights): no vector arithmetic
                const int c = min(c0 + kc, p.Ctot - 1);
                const bool first = c < p.C1;
                const uint32_t soff = first ? st_voff1 + (uint32_t)c * st_cs4 : st_voff2 + (uint32_t)(c - p.C1) * st_cs4;
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int g = sp_goff[(BMODE == BM_SPATIAL) ? i : 0];
                    const uint32_t voff = g >= 0 ? (uint32_t)g * 4u : BUF_OOB;     // loop-invariant: hoisted out of the K loop
                    breg[kc * EPT + i] = first ? bload_s(rx1, voff, soff) : bload_s(rx2, voff, soff);
                }
            }
        }
    };

    // ---------------- registers -> LDS (B gets the BN/activation/gate prologue here) ---------------
    // branch-free per element: parameters come from LDS, validity is a final select (zero padding is
    // applied AFTER the activation: the reference pads activated maps)
    auto commit = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (A4 % NTHREADS == 0 || tid + NTHREADS * i < A4)
                *reinterpret_cast<f32x4*>(As + (tid / (BM / 4) + i * A_ROWS_PER_PASS) * AS + 4 * (tid % (BM / 4))) = areg[i];
        // the prologue kind is launch-uniform: dispatch once, outside the unrolled element loops (a concat
        // of two differently-activated sources falls back to the affine+select form per source)
        const int pro_u = (p.C2 > 0 && p.pro2 != p.pro1) ? -1 : p.pro1;
        auto body = [&](auto tag) {
            constexpr int PRO = decltype(tag)::value;
            if constexpr (BVEC) {
                const int j4 = tid % LPR, kc0 = tid / LPR;
#pragma unroll
                for (int i = 0; i < NBV; ++i) {
                    const int kc = kc0 + i * KSTEPV;
                    const int c = c0 + kc;
                    const int cc = c < p.Ctot ? c : p.Ctot - 1;
                    float sc = 1.0f, sh = 0.0f;
                    if (PRO != S2K_PRO_NONE) { sc = ssc[cc]; sh = ssh[cc]; }
                    f32x4 v = bvec[i];       // no validity select (see the scalar stager)
                    if (PRO != S2K_PRO_NONE || p.gate1) {
                        const float gm = p.gate1 ? gvec[i] : 1.0f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = v[e];
                            if (PRO != S2K_PRO_NONE) t = apply_pro_c<PRO>(t, sc, sh);
                            v[e] = t * gm;
                        }
                    }
                    *reinterpret_cast<f32x4*>(Bs + kc * BN + 4 * j4) = v;
                }
            } else if (BMODE == BM_PIX) {
                constexpr int KSTEP = (NTHREADS >= BN) ? NTHREADS / BN : 1;
                const int j = tid % BN;
                const int kc0 = __builtin_amdgcn_readfirstlane(tid / BN);   // wave-uniform (BN >= 64)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int kc = kc0 + i * KSTEP;
                    const int c = c0 + kc;
                    const int cc = c < p.Ctot ? c : p.Ctot - 1;
                    float v = breg[i];       // no validity select: out-of-range pixels feed discarded output columns,
                    if (PRO != S2K_PRO_NONE) v = apply_pro_c<PRO>(v, ssc[cc], ssh[cc]);   // channels past Ctot meet zero weight rows
                    if (p.gate1) v *= greg[(BMODE == BM_PIX) ? i : 0];
                    Bs[kc * BN + j] = v;
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    const int c = c0 + kc;
                    const bool cok = c < p.Ctot;
                    const int cc = cok ? c : p.Ctot - 1;
                    float sc = 1.0f, sh = 0.0f;
                    if (PRO != S2K_PRO_NONE) { sc = ssc[cc]; sh = ssh[cc]; }
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const int e = tid + NTHREADS * i;
                        float v = breg[kc * EPT + i];      // padding slots were loaded as 0; with a prologue they are zeroed
                        if (PRO != S2K_PRO_NONE)           // again by a 0/1 factor (the reference pads ACTIVATED maps)
                            v = apply_pro_c<PRO>(v, sc, sh) * (sp_goff[(BMODE == BM_SPATIAL) ? i : 0] >= 0 ? 1.0f : 0.0f);
                        if (e < used_sp) Bs[kc * CSB + e] = v;
                    }
                }
            }
        };
        if (pro_u >= 0) {
            dispatch_pro(pro_u, body);
        } else {  // mixed concat (not produced by the planner today): generic per-element form
            if constexpr (BVEC) {
                const int j4 = tid % LPR, kc0 = tid / LPR;
#pragma unroll
                for (int i = 0; i < NBV; ++i) {
                    const int kc = kc0 + i * KSTEPV;
                    const int c = c0 + kc;
                    const bool ok = st_valid && c < p.Ctot;
                    const int cc = c < p.Ctot ? c : p.Ctot - 1;
                    const float gm = p.gate1 ? gvec[i] : 1.0f;
                    f32x4 v = bvec[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? apply_pro(v[e], cc < p.C1 ? p.pro1 : p.pro2, ssc[cc], ssh[cc]) * gm : 0.0f;
                    *reinterpret_cast<f32x4*>(Bs + kc * BN + 4 * j4) = v;
                }
            } else if (BMODE == BM_PIX) {
                constexpr int KSTEP = (NTHREADS >= BN) ? NTHREADS / BN : 1;
                const int j = tid % BN;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int kc = tid / BN + i * KSTEP;
                    const int c = c0 + kc;
                    const int cc = c < p.Ctot ? c : p.Ctot - 1;
                    float v = apply_pro(breg[i], cc < p.C1 ? p.pro1 : p.pro2, ssc[cc], ssh[cc]);
                    if (p.gate1) v *= greg[(BMODE == BM_PIX) ? i : 0];
                    Bs[kc * BN + j] = (st_valid && c < p.Ctot) ? v : 0.0f;
                }
            } else {
                for (int kc = 0; kc < KCH; ++kc) {
                    const int c = c0 + kc;
                    const bool cok = c < p.Ctot;
                    const int cc = cok ? c : p.Ctot - 1;
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const int e = tid + NTHREADS * i;
                        float v = apply_pro(breg[kc * EPT + i], cc < p.C1 ? p.pro1 : p.pro2, ssc[cc], ssh[cc]);
                        v = (cok && sp_goff[(BMODE == BM_SPATIAL) ? i : 0] >= 0) ? v : 0.0f;
                        if (e < used_sp) Bs[kc * CSB + e] = v;
                    }
                }
            }
        }
    };

    const int ch_begin = ksplit * p.chunks_per_split;
    const int ch_end = min(nchunks, ch_begin + p.chunks_per_split);
    __syncthreads();  // ssc / ssh staged
    fetch(ch_begin * KCH);
    commit(ch_begin * KCH);
    __syncthreads();

    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const bool more = ch + 1 < ch_end;
        if (more) fetch((ch + 1) * KCH);
        // ---------------- MFMA over the chunk in LDS --------------------------------------------
        // Explicit two-set operand pipeline: the LDS reads of stage i+1 are issued before the MFMAs of stage i
        // (left alone hipcc reads, waits, multiplies, one k-step at a time).  A stage = G k-steps (>= 4 MFMAs
        // where the tile allows); the last stage of a tap prefetches the first stage of the next tap.
        {
            constexpr int KS = KCH / 2;                          // k-steps (pairs of input channels) per tap
            constexpr int G0 = (WM * WN >= 4) ? 1 : 4 / (WM * WN);
            constexpr int G = (G0 > KS / 2) ? KS / 2 : G0;       // k-steps per stage
            constexpr int SPT = KS / G;                          // stages per tap
            static_assert(KS % G == 0 && SPT % 2 == 0, "stages per tap must be even");
            auto lds_ops = [&](int tap, int toff, int sg, float (&a)[G][WM], float (&b)[G][WN]) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int k = 2 * (sg * G + g) + lh;
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm) a[g][rm] = As[(k * TT + tap) * AS + wm0 + rm * 32 + l31];
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn) b[g][rn] = Bs[k * CSB + boff[rn] + toff];
                }
            };
            auto mfmas = [&](const float (&a)[G][WM], const float (&b)[G][WN]) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                        for (int rn = 0; rn < WN; ++rn)
                            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][rm], b[g][rn], acc[rm][rn], 0, 0, 0);
            };
            float a0[G][WM], b0[G][WN], a1[G][WM], b1[G][WN];
            int tdy = 0, tdx = 0;
            lds_ops(0, 0, 0, a0, b0);
#pragma unroll 1
            for (int tap = 0; tap < TT; ++tap) {
                const int toff = (BMODE == BM_PIX) ? 0 : tdy * p.WS + tdx;
                if (++tdx == p.KW) { tdx = 0; ++tdy; }
                const bool last = tap + 1 == TT;
                const int ntap = last ? tap : tap + 1;           // past the end: re-read (never used)
                const int ntoff = (BMODE == BM_PIX || last) ? toff : tdy * p.WS + tdx;
#pragma unroll
                for (int sg = 0; sg < SPT; sg += 2) {
                    lds_ops(tap, toff, sg + 1, a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (sg + 2 < SPT) lds_ops(tap, toff, sg + 2, a0, b0);
                    else lds_ops(ntap, ntoff, 0, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();  // every wave is done reading this chunk
        if (more) {
            commit((ch + 1) * KCH);
            __syncthreads();
        }
    }

    // ---------------- epilogue ---------------------------------------------------------------------
    if (p.splits > 1) {   // partial tile -> scratch; bias / residual / statistics are applied by splitk_reduce_kernel
        float* part = p.scratch + (int64_t)ksplit * p.y_elems;
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
    if (scatter) {
        // rows m = (co,dy,dx): registers 4q..4q+3 of a lane are the 2x2 output patch of one co
        const int64_t plane = 4 * (int64_t)p.HW;
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = wm0 + rm * 32 + 8 * q + 4 * lh;
                const int gm = m0 + row;
                if (gm < p.M) {
                    const int co = gm >> 2;
                    const float bs = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        if (cval[rn]) {
                            float* dst = p.y + ycol[rn] + (int64_t)co * plane;
                            float2 r0 = make_float2(acc[rm][rn][4 * q + 0] + bs, acc[rm][rn][4 * q + 1] + bs);
                            float2 r1 = make_float2(acc[rm][rn][4 * q + 2] + bs, acc[rm][rn][4 * q + 3] + bs);
                            *reinterpret_cast<float2*>(dst) = r0;
                            *reinterpret_cast<float2*>(dst + 2 * p.W) = r1;
                        }
                }
            }
        return;
    }
    if constexpr (BN >= 64 && BN <= 256) {
        if (p.tepi) {
            // ---- transposed store (see conv_bf16.hip): the tile goes through LDS, 32 rows per wave row at a time, and comes back
            // pixel-major - a lane stores 4 consecutive pixels (16 bytes), an instruction rows x 512 bytes instead of 2 rows x 128 bytes
            // straight from the MFMA layout (write-heavy layers streamed their output at ~3 TB/s where a fill reaches 6.8).  Bias /
            // residual / accumulate and the BatchNorm statistics are applied on the reading side (a row's sums: one DPP reduction over
            // the lanes that hold it, one f64 atomic pair per row and workgroup, fixed order throughout).
            constexpr int CTP = BN + 8;                          // row stride: lane halves (rows r, r + 4) land in different banks
            constexpr int PASS_ROWS = WVM * 32;
            constexpr int G = BN / 4, RPI = NTHREADS / G;
            static_assert(NTHREADS % G == 0 && PASS_ROWS % RPI == 0 && (G == 16 || G == 32 || G == 64), "transposed epilogue geometry");
            const int g4 = tid % G, r0 = tid / G;
            bool gok;
            int64_t gcol;
            const int j = 4 * g4;
            if (BMODE == BM_PIX) {
                const int n = nt * BN + j;
                gok = n < p.Ntot;
                const int nn = gok ? n : 0;
                const int b = nn / p.HW, pp = nn - b * p.HW;
                gcol = (int64_t)b * p.YC * HWo + pp;
            } else {
                const int tx = nt % p.tiles_x, ty = (nt / p.tiles_x) % p.tiles_y, sb = nt / (p.tiles_x * p.tiles_y);
                const int r = j / p.XW, xx = j - r * p.XW;
                gok = (r < p.R) && (ty * p.R + r < p.HO) && (tx * p.XW + xx < p.WO);
                gcol = (int64_t)sb * p.YC * HWo + (int64_t)(ty * p.R + r) * p.WO + (tx * p.XW + xx);
            }
            double* st = p.stats ? p.stats + (int64_t)(tile % p.nrep) * 2 * p.M : nullptr;
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) {
                if (rm) __syncthreads();                         // the previous pass has been read
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rl = (wave / WVN) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn) smem[rl * CTP + wn0 + rn * 32 + l31] = acc[rm][rn][reg];
                }
                __syncthreads();
#pragma unroll
                for (int r = r0; r < PASS_ROWS; r += RPI) {
                    const int gm = m0 + (r >> 5) * (WM * 32) + rm * 32 + (r & 31);
                    const bool ok = gok && gm < p.M;
                    f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * CTP + 4 * g4);
                    float s = 0.0f, q = 0.0f;
                    if (ok) {
                        if (p.bias) { const float bsv = p.bias[gm]; v[0] += bsv; v[1] += bsv; v[2] += bsv; v[3] += bsv; }
                        float* dst = p.y + gcol + (int64_t)gm * HWo;
                        if (p.res) { const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + gcol + (int64_t)gm * HWo); v[0] = res_combine(v[0], rv[0], p.res_mul); v[1] = res_combine(v[1], rv[1], p.res_mul); v[2] = res_combine(v[2], rv[2], p.res_mul); v[3] = res_combine(v[3], rv[3], p.res_mul); }
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
    }
    // row sums for BatchNorm: lanes -> half-wave shuffle, waves -> LDS slots summed in a FIXED order (no
    // float atomics: BatchNorm statistics stay run-to-run reproducible), then ONE f64 atomic pair per row
    // per workgroup into the statistics replica of this tile
    float* srow = smem;  // [WVN][2][BM], the K loop is over: LDS is free
    const int wn_idx = wave % WVN;
#pragma unroll
    for (int rm = 0; rm < WM; ++rm)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const int gm = m0 + row;
            const bool rok = gm < p.M;
            const float bs = (p.bias && rok) ? p.bias[gm] : 0.0f;
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) {
                float v = acc[rm][rn][reg] + bs;
                if (rok && cval[rn]) {
                    float* dst = p.y + ycol[rn] + (int64_t)gm * HWo;
                    if (p.res) v = res_combine(v, p.res[ycol[rn] + (int64_t)gm * HWo], p.res_mul);
                    if (p.beta) v += *dst;
                    *dst = v;
                    s += v;
                    q += v * v;
                }
            }
            if (p.stats) {
                s = half_sum_hi(s);
                q = half_sum_hi(q);
                if (l31 == 31) {  // each (wave-column, row) slot has exactly one writer
                    srow[(wn_idx * 2 + 0) * BM + row] = rok ? s : 0.0f;
                    srow[(wn_idx * 2 + 1) * BM + row] = rok ? q : 0.0f;
                }
            }
        }
    if (p.stats) {
        __syncthreads();
        double* st = p.stats + (int64_t)(tile % p.nrep) * 2 * p.M;
        for (int i = tid; i < 2 * BM; i += NTHREADS) {
            const int row = i % BM, which = i / BM, gm = m0 + row;
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < WVN; ++w) tot += srow[(w * 2 + which) * BM + row];
            if (gm < p.M) atomic_add_d(st + which * p.M + gm, (double)tot);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
template <typename T>
static T* ref_ptr(const Ctx& c, int64_t ref) {
    if (ref < 0) return nullptr;
    const int base = (int)(ref >> 56);
    const int64_t off = ref & ((1ll << 56) - 1);
    if (base >= c.n_bases || c.bases[base] == nullptr) return reinterpret_cast<T*>(1);  // flagged by caller
    return reinterpret_cast<T*>(static_cast<char*>(c.bases[base]) + off);
}

static int pick_bm(int M) {
    // largest BM in {128, 64, 32} wasting <= 12 % of the rows, else the least wasteful
    const int cands[3] = {128, 64, 32};
    int best = 32;
    double best_w = 1e9;
    for (int bm : cands) {
        const double w = (double)cdiv(M, bm) * bm / M - 1.0;
        if (w <= 0.12) return bm;
        if (w < best_w - 1e-9) { best_w = w; best = bm; }
    }
    return best;
}

// Split-K tail: Y[b][m][hw] (+)= sum_s part[s][b][m][hw] + bias[m] + res, partials added in a FIXED order (reproducible),
// BatchNorm statistics of the finished rows.  One wave per (b, m) plane.
__global__ void __launch_bounds__(NTHREADS) splitk_reduce_kernel(const ConvP p) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int HWo = p.HO * p.WO;
    if (plane >= (int64_t)p.B * p.M) return;
    const int b = (int)(plane / p.M), m = (int)(plane % p.M);
    const int64_t base = ((int64_t)b * p.YC + m) * HWo;
    const float bs = p.bias ? p.bias[m] : 0.0f;
    float s = 0.0f, q = 0.0f;
    for (int i = lane; i < HWo; i += 64) {
        float v = bs;
        for (int k = 0; k < p.splits; ++k) v += p.scratch[(int64_t)k * p.y_elems + base + i];
        if (p.res) v = res_combine(v, p.res[base + i], p.res_mul);
        if (p.beta) v += p.y[base + i];
        p.y[base + i] = v;
        s += v;
        q = fmaf(v, v, q);
    }
    if (p.stats) {
        s = wave_sum_hi(s);
        q = wave_sum_hi(q);
        if (lane == 63) {
            double* st = p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.M;
            atomic_add_d(st + m, (double)s);
            atomic_add_d(st + p.M + m, (double)q);
        }
    }
}

void launch_splitk_reduce(const ConvP& p, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64((int64_t)p.B * p.M, 4)), dim3(NTHREADS), 0, st, p);
}

template <int BMODE, int TT, int WM, int WN, int WVM, int WVN, int KCH, int EPT, int MINW = 2, bool BVEC = false>
static int launch_cfg(ConvP& p, int n_ntiles, hipStream_t st, bool allow_splitk = false) {
    constexpr int BM = WM * WVM * 32, BN = WN * WVN * 32;
    constexpr int A_FLOATS = (KCH * TT * BM + 3) & ~3;
    p.n_mtiles = cdiv(p.M, BM);
    const size_t b_floats = (BMODE == BM_PIX) ? (size_t)KCH * BN : (size_t)KCH * NTHREADS * EPT;
    p.b_floats = (int)b_floats;
    const size_t lds = (A_FLOATS + b_floats + 2 * (size_t)p.Ctot) * sizeof(float);
    if (BMODE == BM_SPATIAL && p.gate1) { set_error("conv: SE gate is only supported on 1x1 convs"); return S2K_EINVAL; }
    auto kern = conv_igemm_kernel<BMODE, TT, WM, WN, WVM, WVN, KCH, EPT, MINW, BVEC>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (lds > 160 * 1024) { set_error("conv: LDS %zu too large", lds); return S2K_EINVAL; }
    {   // 32-bit buffer offsets: image-local tiles need one image < 2 GiB, tiles that may straddle images the whole tensor
        // (descriptors are based at the first image a tile touches)
        const int64_t span = (BMODE == BM_SPATIAL || (p.HW % BN) == 0) ? 1 : std::min<int64_t>(p.B, (BN - 2) / p.HW + 2);
        const int64_t img1 = (int64_t)p.C1 * p.H * p.W * 4, img2 = (int64_t)p.C2 * p.H * p.W * 4;
        const int64_t need = std::max(img1, img2) * span;
        if (need >= 0x7ffffff0ll) { set_error("conv: the %lld image(s) one tile touches exceed 2 GiB (%lld B)", (long long)span, (long long)need); return S2K_EINVAL; }
    }
    if (BMODE == BM_SPATIAL && p.IR * p.WS > NTHREADS * EPT) { set_error("conv: halo tile exceeds EPT"); return S2K_EINVAL; }
    const int64_t blocks = (int64_t)p.n_mtiles * n_ntiles;
    if (blocks <= 0 || blocks > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)blocks); return S2K_EINVAL; }
    p.n_tiles = (int)blocks;
    const int nchunks = cdiv(p.Ctot, KCH);
    p.splits = 1;
    p.chunks_per_split = nchunks;
    p.y_elems = (int64_t)p.B * p.YC * p.HO * p.WO;
    static const int force_splits = tune_int("S2K_SPLITS", 0);
    if (force_splits > 1 && allow_splitk && p.scratch && p.mode == S2K_MODE_CONV && nchunks >= 2 * force_splits) {
        p.chunks_per_split = cdiv(nchunks, force_splits);
        p.splits = cdiv(nchunks, p.chunks_per_split);
    } else if (force_splits == 0 && allow_splitk && p.scratch && p.mode == S2K_MODE_CONV && nchunks >= 4) {
        // too few tiles to fill 256 CUs and a long reduction: cut K so that several workgroups per CU hide each other's
        // latency (measured on 64x64 tiles: 160..512 tiles want ~1024 workgroups, 600 tiles x 48 chunks want x3..4)
        int splits = 1;
        if (blocks <= 512) splits = (int)cdiv64(1024, blocks);
        else if (blocks <= 1024 && nchunks >= 32) splits = (int)cdiv64(1536, blocks);
        if (splits > 8) splits = 8;
        if (splits > nchunks / 2) splits = nchunks / 2;
        if (splits > 1) {
            p.chunks_per_split = cdiv(nchunks, splits);
            p.splits = cdiv(nchunks, p.chunks_per_split);
        }
    }
    // transposed store of the output tile (whole tiles only, pixel quads aligned to 16 bytes)
    size_t lds_run = lds;
    p.tepi = 0;
    if (BN >= 64 && BN <= 256 && p.splits == 1 && p.mode != S2K_MODE_CONVT_SCATTER &&
        ((BMODE == BM_PIX && (p.HW & 3) == 0) || (BMODE == BM_SPATIAL && (p.XW & 3) == 0 && (p.WO & 3) == 0))) {
        p.tepi = 1;
        lds_run = std::max(lds, (size_t)WVM * 32 * (BN + 8) * sizeof(float));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.splits), dim3(NTHREADS), lds_run, st, p);
    if (p.splits > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64((int64_t)p.B * p.M, 4)), dim3(NTHREADS), 0, st, p);
    return S2K_OK;
}

// tile geometry for the SPATIAL stager; BN = output pixels per tile, cap = halo floats per channel
static bool spatial_tiling(ConvP& p, int BN, int cap) {
    p.XW = p.WO <= BN ? p.WO : BN;
    int R = BN / p.XW;
    if (R > p.HO) R = p.HO;
    if (R < 1) R = 1;
    for (;; --R) {
        p.R = R;
        p.IR = (R - 1) * p.S + p.KH;
        p.IC = (p.XW - 1) * p.S + p.KW;
        p.WS = p.IC;
        if (p.IR * p.WS <= cap) break;
        if (R == 1) return false;
    }
    p.CS = p.IR * p.WS;
    p.tiles_x = cdiv(p.WO, p.XW);
    p.tiles_y = cdiv(p.HO, p.R);
    return true;
}

int launch_conv(const S2kOp& op, const Ctx& c) {
    ConvP p{};
    p.x1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_X1]);
    p.bnv1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_BNV1]);
    p.gate1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_GATE1]);
    p.x2 = ref_ptr<const float>(c, op.t[S2K_CONV_T_X2]);
    p.bnv2 = ref_ptr<const float>(c, op.t[S2K_CONV_T_BNV2]);
    p.wt = ref_ptr<const float>(c, op.t[S2K_CONV_T_WT]);
    p.bias = ref_ptr<const float>(c, op.t[S2K_CONV_T_BIAS]);
    p.y = ref_ptr<float>(c, op.t[S2K_CONV_T_Y]);
    p.stats = ref_ptr<double>(c, op.t[S2K_CONV_T_STATS]);
    p.res = ref_ptr<const float>(c, op.t[S2K_CONV_T_RES]);
    p.scratch = ref_ptr<float>(c, op.t[S2K_CONV_T_SCRATCH]);
    p.wtb = (op.flags & S2K_FLAG_BF16) ? ref_ptr<const void>(c, op.t[S2K_CONV_T_WTB]) : nullptr;
    p.wtq = (!(op.flags & S2K_FLAG_BF16) && (op.flags & S2K_FLAG_Q4)) ? ref_ptr<const float>(c, op.t[S2K_CONV_T_WTB]) : nullptr;
    p.force_dma = (op.flags & S2K_FLAG_DMA) ? 1 : 0;
    p.res_mul = (op.flags & S2K_FLAG_RES_GELU_GRAD) ? S2K_PRO_GELU : 0;
    if (p.res_mul && (!p.res || (op.flags & S2K_FLAG_BF16))) { set_error("conv: S2K_FLAG_RES_GELU_GRAD needs RES and an f32 stage"); return S2K_EINVAL; }
    static const int cv_exp = tune_int("S2K_CV_EXP", 0);
    p.exp = cv_exp;
    const void* ptrs[] = {p.x1, p.bnv1, p.gate1, p.x2, p.bnv2, p.wt, p.bias, p.y, p.stats, p.res, p.scratch, p.wtb, p.wtq};
    for (const void* q : ptrs)
        if (q == reinterpret_cast<const void*>(1)) { set_error("conv: tensor references a null base"); return S2K_EFAULT; }
    const int32_t* d = op.d;
    p.B = d[S2K_CONV_D_B]; p.C1 = d[S2K_CONV_D_C1]; p.C2 = d[S2K_CONV_D_C2];
    p.H = d[S2K_CONV_D_H]; p.W = d[S2K_CONV_D_W]; p.M = d[S2K_CONV_D_M];
    p.KH = d[S2K_CONV_D_KH]; p.KW = d[S2K_CONV_D_KW]; p.S = d[S2K_CONV_D_STRIDE];
    p.PT = d[S2K_CONV_D_PAD_T]; p.PL = d[S2K_CONV_D_PAD_L]; p.HO = d[S2K_CONV_D_HO]; p.WO = d[S2K_CONV_D_WO];
    p.pro1 = d[S2K_CONV_D_PRO1]; p.pro2 = d[S2K_CONV_D_PRO2]; p.mode = d[S2K_CONV_D_MODE];
    p.w_sm = d[S2K_CONV_D_W_SM]; p.w_sk = d[S2K_CONV_D_W_SK]; p.w_st = d[S2K_CONV_D_W_ST];
    p.flip = d[S2K_CONV_D_FLIP]; p.beta = d[S2K_CONV_D_BETA]; p.YC = d[S2K_CONV_D_YC];
    p.nrep = d[S2K_CONV_D_NREP] > 0 ? d[S2K_CONV_D_NREP] : 1;
    p.x1_bf16 = d[S2K_CONV_D_X1_BF16];
    const int T = p.KH * p.KW;
    p.Ctot = p.C1 + p.C2;
    p.HW = p.H * p.W;
    p.a_mfast = 1;
    if (p.w_sm != 1 || p.flip || (p.w_st & 3) || p.w_sk != T * p.w_st || p.w_st < ((p.M + 127) / 128) * 128) {
        set_error("conv: weights must be in the WEIGHT_PACK layout (W_SM=1, W_ST=MP, W_SK=T*MP, FLIP=0)");
        return S2K_EINVAL;
    }
    p.R = p.XW = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = p.CS = 0;
    p.n_tiles = p.n_mtiles = 0;
    if (!p.x1 || !p.wt || !p.y || p.B <= 0 || p.C1 <= 0 || p.M <= 0 || p.H <= 0 || p.W <= 0) {
        set_error("conv: missing tensor or non-positive dimension");
        return S2K_EINVAL;
    }
    if (p.C2 > 0 && !p.x2) { set_error("conv: C2 > 0 without X2"); return S2K_EINVAL; }
    if ((p.pro1 != S2K_PRO_NONE && !p.bnv1) || (p.C2 > 0 && p.pro2 != S2K_PRO_NONE && !p.bnv2)) {
        set_error("conv: prologue without BNV"); return S2K_EINVAL;
    }
    const bool pix = (T == 1 && p.S == 1) || p.mode != S2K_MODE_CONV;
    if (p.mode != S2K_MODE_CONV) {
        if (T != 1 || p.S != 1 || p.C2 != 0 || p.HO != p.H || p.WO != p.W) {
            set_error("conv: scatter/gather modes are 1x1 over the low-resolution grid"); return S2K_EINVAL;
        }
        if (p.mode == S2K_MODE_CONVT_SCATTER && ((p.M & 3) || p.beta || p.stats || p.res)) {
            set_error("conv: scatter needs M = 4*Cout, beta = 0, no stats, no residual"); return S2K_EINVAL;
        }
        if (p.mode == S2K_MODE_GATHER2X2 && ((p.C1 & 3) || p.pro1 != S2K_PRO_NONE || p.gate1)) {
            set_error("conv: gather needs C1 = 4*Cout and no prologue"); return S2K_EINVAL;
        }
    }
    if (pix && (p.HO != p.H || p.WO != p.W || p.PT || p.PL)) { set_error("conv: 1x1 geometry mismatch"); return S2K_EINVAL; }
    const int64_t ntot = (int64_t)p.B * p.HW;
    if (ntot > 0x7fffffff) { set_error("conv: too many pixels"); return S2K_EINVAL; }
    p.Ntot = (int)ntot;

    hipStream_t st = c.stream;
    if (p.wtb) {   // bf16-mixed plan: the shapes of conv_bf16.hip round their MFMA operands to bf16; 1 = not one of its shapes
        const int rc = launch_conv_bf16(p, st);
        if (rc != 1) return rc;
        p.R = p.XW = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = p.CS = 0;
        p.n_tiles = p.n_mtiles = 0;
    }
    if (p.x1_bf16) {   // planned only where conv_bf16.hip takes the stage (plan/bf16.py); the f32 kernels would read the halves as floats
        set_error("conv: X1_BF16 on a stage the bf16 1x1 kernel does not take (FLAG_BF16 missing or shape not in its list)");
        return S2K_EINVAL;
    }
    if (p.wtq) {   // prologue-free 1x1 contractions with the quad weight copy: conv_q4.hip where its launcher's measured routing rule says
                   // so (S2K_FLAG_DMA beside S2K_FLAG_Q4: every shape it supports - tests); 1 = not one of its shapes
        const int rc = launch_conv_q4(p, st);
        if (rc != 1) return rc;
        p.n_tiles = p.n_mtiles = 0;
        p.splits = 1;
    } else if (p.force_dma) {
        const int rc = launch_conv_dma(p, st);
        if (rc != 1) return rc;
        p.n_tiles = p.n_mtiles = 0;
        p.splits = 1;
    }
    {   // the prologue-light, MFMA-bound shapes run on the producer / consumer kernels (igemm_pc.hip); 1 = not one of theirs
        const int rc = launch_conv_pc(p, st);
        if (rc != 1) return rc;
        p.R = p.XW = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = p.CS = 0;
        p.n_tiles = p.n_mtiles = 0;
    }
    {   // prologue-free 1x1 contractions the producer / consumer kernel leaves (short or deep reductions, few pixels): the LDS-DMA
        // ring kernel of conv_dma.hip; 1 = not one of its shapes
        const int rc = launch_conv_dma(p, st);
        if (rc != 1) return rc;
        p.n_tiles = p.n_mtiles = 0;
        p.splits = 1;
    }
    const int bm = pick_bm(p.M);
    if (pix) {
        // small problems (deep 8x8 / 16x16 maps): 64x64 tiles keep more CUs busy
        const int64_t tiles_big = (int64_t)cdiv(p.M, bm) * cdiv(p.Ntot, bm == 128 ? 128 : 256);
        static const int small_max = tune_int("S2K_PIX_SMALL_TILES", 400);
        static const int k16_max = tune_int("S2K_PIX_K16_MAX", 192);
        static const int novec = tune_int("S2K_PIX_NOVEC", 0);
        const bool bvec = !novec && p.mode != S2K_MODE_GATHER2X2 && (p.HW & 3) == 0;
#define PIX_CFG(WMv, WNv, WVMv, WVNv, KCHv, ntl, sk) \
    (bvec ? launch_cfg<BM_PIX, 1, WMv, WNv, WVMv, WVNv, KCHv, 1, 2, true>(p, ntl, st, sk) \
          : launch_cfg<BM_PIX, 1, WMv, WNv, WVMv, WVNv, KCHv, 1, 2, false>(p, ntl, st, sk))
        static const int force_cfg = tune_int("S2K_PIX_FORCE", 0);
        if (force_cfg == 1) return PIX_CFG(1, 1, 2, 2, 64, cdiv(p.Ntot, 64), true);
        if (force_cfg == 2) return PIX_CFG(2, 2, 2, 2, 64, cdiv(p.Ntot, 128), true);
        if (force_cfg == 3) return PIX_CFG(2, 2, 1, 4, 16, cdiv(p.Ntot, 256), false);
        if (force_cfg == 4) return PIX_CFG(1, 2, 1, 4, 16, cdiv(p.Ntot, 256), false);
        if (force_cfg == 5) return PIX_CFG(2, 2, 2, 2, 16, cdiv(p.Ntot, 128), false);
        if (bm >= 64 && tiles_big < small_max) return PIX_CFG(1, 1, 2, 2, 64, cdiv(p.Ntot, 64), true);
        if (bm == 128 && p.Ctot <= k16_max) return PIX_CFG(2, 2, 2, 2, 16, cdiv(p.Ntot, 128), false);   // short-K expand convs
        if (bm == 128) return PIX_CFG(2, 2, 2, 2, 64, cdiv(p.Ntot, 128), false);
        if (bm == 64) return PIX_CFG(2, 2, 1, 4, 16, cdiv(p.Ntot, 256), false);
        return PIX_CFG(1, 2, 1, 4, 16, cdiv(p.Ntot, 256), false);
#undef PIX_CFG
    }
    if (T != 9 || p.S > 2) { set_error("conv: unsupported kernel %dx%d stride %d", p.KH, p.KW, p.S); return S2K_EINVAL; }
    int64_t tiles = 0;
    auto geom = [&](int BN, int cap) -> bool {
        if (!spatial_tiling(p, BN, cap)) return false;
        tiles = (int64_t)p.B * p.tiles_x * p.tiles_y;
        return true;
    };
    const int64_t out_px = (int64_t)p.B * p.HO * p.WO;
    if (bm == 128) {
        if (!geom(128, 2 * NTHREADS)) { set_error("conv: halo tile does not fit (W=%d)", p.W); return S2K_EINVAL; }
        if (tiles * cdiv(p.M, 128) >= 160) return launch_cfg<BM_SPATIAL, 9, 2, 2, 2, 2, 8, 2>(p, (int)tiles, st);
    }
    if (bm >= 64) {
        if ((int64_t)cdiv(p.M, 64) * cdiv64(out_px, 256) >= 160 && geom(256, 4 * NTHREADS))
            return launch_cfg<BM_SPATIAL, 9, 2, 2, 1, 4, 8, 4>(p, (int)tiles, st);
        if (!geom(64, 2 * NTHREADS)) { set_error("conv: halo tile does not fit"); return S2K_EINVAL; }
        return launch_cfg<BM_SPATIAL, 9, 1, 1, 2, 2, 8, 2>(p, (int)tiles, st);
    }
    // thin layers (M <= 32, full-resolution maps): 32 x 1024 tiles amortise the halo (6 rows per 4)
    static const int thin_mode = tune_int("S2K_THIN", 1);
    if (thin_mode == 1 && p.S == 1 && cdiv64(out_px, 512) >= 200 && geom(512, 5 * NTHREADS))
        return launch_cfg<BM_SPATIAL, 9, 1, 4, 1, 4, 8, 5>(p, (int)tiles, st);
    if (p.S == 1 && cdiv64(out_px, 1024) >= 200 && geom(1024, 7 * NTHREADS))
        return launch_cfg<BM_SPATIAL, 9, 1, 8, 1, 4, 8, 7, 1>(p, (int)tiles, st);
    if (!geom(256, 4 * NTHREADS)) { set_error("conv: halo tile does not fit"); return S2K_EINVAL; }
    return launch_cfg<BM_SPATIAL, 9, 1, 2, 1, 4, 8, 4>(p, (int)tiles, st);
}

// ---------------------------------------------------------------------------------------------
// lane-map self test (exact integer data): D = A[32x2] * B[2x32]
// ---------------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l31 * 2 + lh], b[lh * 32 + l31], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}

int launch_mfma_selftest(const float* a, const float* b, float* d, hipStream_t st) {
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, st, a, b, d);
    return S2K_OK;
}

}  // namespace s2k
