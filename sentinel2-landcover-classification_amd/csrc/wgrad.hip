// Weight gradients of the dense convs on the gfx950 f32 matrix cores; K = pixels.
//
// Replaces ATen convolution_backward's grad_weight for nn.Conv2d 1x1 / 3x3 / stem and
// nn.ConvTranspose2d k2 s2 (reference efficientnet_unet.py:114-121,168-176,191-197,319-372):
//   conv :  dW[tap][m][c] += sum_pix dY[b][m][yo][xo]      * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
//   convT:  dW[tap][m][c] += sum_pix Xpro[b][m][y][x]      * G[b][c][2y+dy][2x+dx]
// MFMA 32x32x2 with k = two consecutive pixels: lane l holds P[m = l&31][pix + (l>>5)] and
// Q[c = l&31][pix + (l>>5) (+tap shift)]; both LDS tiles are [channel][pixels] with an ODD channel
// stride, so the 32 lanes of a half-wave hit 32 different banks.  One accumulator tile per tap; the
// activations are re-normalised/activated on load (same prologue as the forward), never stored.
// Partial sums of the pixel splits are combined with float atomics into a [T][M][C] scratch whose
// rows are contiguous along c (128-B segments per half-wave: the full-rate atomic shape).
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
};

__device__ __forceinline__ float ld_pro(const float* x, const float* bnv, const float* gate, int pro, int C, int c,
                                        int64_t off, int gate_row) {
    float v = x[off];
    if (pro != S2K_PRO_NONE) v = apply_pro(v, pro, bnv[c], bnv[C + c]);
    if (gate) v *= gate[gate_row + c];
    return v;
}

// T taps, per-wave tile (WM*32) x (WN*32), waves arranged WVM x WVN x WVK (K = pixel pairs)
template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK>
__global__ void __launch_bounds__(NTHREADS) wgrad_kernel(const WgradP p) {
    constexpr int BM = WM * WVM * 32;
    constexpr int BC = WN * WVN * 32;
    static_assert(WVM * WVN * WVK == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ps = smem;                    // [BM][PSTR]
    float* Qs = smem + BM * p.PSTR;      // [BC][CSQ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wk = wave % WVK;
    const int wmn = wave / WVK;
    const int wm0 = (wmn / WVN) * (WM * 32);
    const int wc0 = (wmn % WVN) * (WN * 32);
    const int mt = blockIdx.x % p.n_mtiles;
    const int ct = blockIdx.x / p.n_mtiles;
    const int m0 = mt * BM, c0 = ct * BC;
    const int tile_begin = blockIdx.y * p.tiles_per_split;
    int tile_end = tile_begin + p.tiles_per_split;
    if (tile_end > p.ntiles) tile_end = p.ntiles;

    f32x16 acc[T][WM][WN];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.0f;

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();
        int npairs;  // pixel pairs in this tile (k steps)
        if (MODE == WG_SPATIAL) {
            const int tx = tile % p.tiles_x;
            const int ty = (tile / p.tiles_x) % p.tiles_y;
            const int b = tile / (p.tiles_x * p.tiles_y);
            const int y0 = ty * p.R, x0 = tx * p.XW;
            // P tile: Ps[m][r*XWe + xx], zero outside the image / beyond M
            const int np = p.R * p.XWe;
            for (int idx = tid; idx < BM * np; idx += NTHREADS) {
                const int m = idx / np, j = idx - m * np;
                const int r = j / p.XWe, xx = j - r * p.XWe;
                const int gm = m0 + m, yo = y0 + r, xo = x0 + xx;
                float v = 0.0f;
                if (gm < p.M && xx < p.XW && yo < p.HO && xo < p.WO)
                    v = ld_pro(p.p, p.bnvp, p.gatep, p.prop, p.M, gm,
                               ((int64_t)b * p.M + gm) * p.HWp + (int64_t)yo * p.WO + xo, b * p.M);
                Ps[m * p.PSTR + j] = v;
            }
            // Q halo tile: Qs[c][rr*WS + cc]
            const int iy0 = y0 * p.S - p.PT, ix0 = x0 * p.S - p.PL;
            const int used = p.IR * p.WS;
            int goff[WG_EPT_MAX];
#pragma unroll
            for (int i = 0; i < WG_EPT_MAX; ++i) {
                const int e = tid + NTHREADS * i;
                int g = -1;
                if (e < used) {
                    const int rr = e / p.WS, cc = e - rr * p.WS;
                    const int iy = iy0 + rr, ix = ix0 + cc;
                    if (cc < p.IC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) g = iy * p.W + ix;
                }
                goff[i] = g;
            }
            for (int c = 0; c < BC; ++c) {
                const int gc = c0 + c;
                const bool cok = gc < p.C;
                const int64_t plane = ((int64_t)b * p.C + gc) * p.HWq;
#pragma unroll
                for (int i = 0; i < WG_EPT_MAX; ++i) {
                    const int e = tid + NTHREADS * i;
                    if (e < used) {
                        float v = 0.0f;
                        if (cok && goff[i] >= 0)
                            v = ld_pro(p.q, p.bnvq, p.gateq, p.proq, p.C, gc, plane + goff[i], b * p.C);
                        Qs[c * p.CSQ + e] = v;
                    }
                }
            }
            npairs = np >> 1;
        } else {
            // flat pixel chunk [n0, n0 + NP) over (b, y, x) of the P-resolution grid
            const int64_t ntot = (int64_t)p.B * p.HWp;
            const int64_t n0 = (int64_t)tile * p.NP;
            const int j = tid % p.NP, rq = tid / p.NP, rstep = NTHREADS / p.NP;
            const int64_t n = n0 + j;
            const bool ok = n < ntot;
            const int64_t nn = ok ? n : 0;
            const int b = (int)(nn / p.HWp);
            const int pp = (int)(nn - (int64_t)b * p.HWp);
            for (int m = rq; m < BM; m += rstep) {
                const int gm = m0 + m;
                float v = 0.0f;
                if (ok && gm < p.M)
                    v = ld_pro(p.p, p.bnvp, p.gatep, p.prop, p.M, gm, ((int64_t)b * p.M + gm) * p.HWp + pp, b * p.M);
                Ps[m * p.PSTR + j] = v;
            }
            if (MODE == WG_PIX) {
                for (int c = rq; c < BC; c += rstep) {
                    const int gc = c0 + c;
                    float v = 0.0f;
                    if (ok && gc < p.C)
                        v = ld_pro(p.q, p.bnvq, p.gateq, p.proq, p.C, gc, ((int64_t)b * p.C + gc) * p.HWq + pp, b * p.C);
                    Qs[c * p.CSQ + j] = v;
                }
            } else {  // GATHER: Q is [B][C][2HO][2WO]; tap (dy,dx) plane t at Qs[c][t*NP + j]
                const int yy = pp / p.WO, xx = pp - yy * p.WO;
                const int64_t g0 = (int64_t)(2 * yy) * p.W + 2 * xx;
                for (int c = rq; c < BC; c += rstep) {
                    const int gc = c0 + c;
                    float2 r0 = make_float2(0.f, 0.f), r1 = make_float2(0.f, 0.f);
                    if (ok && gc < p.C) {
                        const float* src = p.q + ((int64_t)b * p.C + gc) * p.HWq + g0;
                        r0 = *reinterpret_cast<const float2*>(src);
                        r1 = *reinterpret_cast<const float2*>(src + p.W);
                    }
                    float* dst = Qs + c * p.CSQ + j;
                    dst[0] = r0.x;
                    dst[p.NP] = r0.y;
                    dst[2 * p.NP] = r1.x;
                    dst[3 * p.NP] = r1.y;
                }
            }
            npairs = p.NP >> 1;
        }
        __syncthreads();

        // ---------------- MFMA over pixel pairs ----------------------------------------------------
        for (int s = wk; s < npairs; s += WVK) {
            const int n = 2 * s + lh;  // this lane's pixel inside the tile
            float a[WM];
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = Ps[(wm0 + rm * 32 + l31) * p.PSTR + n];
            int qbase;
            if (MODE == WG_SPATIAL) {
                const int r = n / p.XWe, xx = n - r * p.XWe;
                qbase = (r * p.S) * p.WS + xx * p.S;
            } else {
                qbase = n;
            }
            int tdy = 0, tdx = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int toff;
                if (MODE == WG_SPATIAL) toff = tdy * p.WS + tdx;
                else if (MODE == WG_GATHER) toff = t * p.NP;
                else toff = 0;
                float bq[WN];
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) bq[rn] = Qs[(wc0 + rn * 32 + l31) * p.CSQ + qbase + toff];
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        acc[t][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm], bq[rn], acc[t][rm][rn], 0, 0, 0);
                if (++tdx == p.KW) { tdx = 0; ++tdy; }
            }
        }
    }

    // ---------------- combine: wgs[t][m][c] += acc ------------------------------------------------
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) {
                const int gc = c0 + wc0 + rn * 32 + l31;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    if (gm < p.M && gc < p.C)
                        atomicAdd(p.wgs + ((int64_t)t * p.M + gm) * p.CTOT + gc, acc[t][rm][rn][reg]);
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

template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK>
static int launch_wg(WgradP& p, hipStream_t st) {
    constexpr int BM = WM * WVM * 32, BC = WN * WVN * 32;
    p.n_mtiles = cdiv(p.M, BM);
    p.n_ctiles = cdiv(p.C, BC);
    const size_t lds = ((size_t)BM * p.PSTR + (size_t)BC * p.CSQ) * sizeof(float);
    if (lds > 160 * 1024) { set_error("wgrad: LDS %zu too large", lds); return S2K_EINVAL; }
    auto kern = wgrad_kernel<MODE, T, WM, WN, WVM, WVN, WVK>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    // pixel splits: aim at ~3 workgroups per CU, at least 4 tiles per split to amortise the combine
    const int mc = p.n_mtiles * p.n_ctiles;
    int splits = cdiv(768, mc);
    const int max_splits = cdiv(p.ntiles, 4);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    p.tiles_per_split = cdiv(p.ntiles, splits);
    splits = cdiv(p.ntiles, p.tiles_per_split);
    hipLaunchKernelGGL(kern, dim3(mc, splits), dim3(NTHREADS), lds, st, p);
    return S2K_OK;
}

int launch_wgrad(const S2kOp& op, const Ctx& c) {
    WgradP p;
    p.p = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_P]);
    p.bnvp = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_BNVP]);
    p.gatep = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_GATEP]);
    p.q = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_Q]);
    p.bnvq = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_BNVQ]);
    p.gateq = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_GATEQ]);
    p.wgs = ref_ptr<float>(c, op.t[S2K_WGRAD_T_WGS]);
    const void* ptrs[] = {p.p, p.bnvp, p.gatep, p.q, p.bnvq, p.gateq, p.wgs};
    for (const void* q : ptrs)
        if (q == reinterpret_cast<const void*>(1)) { set_error("wgrad: tensor references a null base"); return S2K_EFAULT; }
    const int32_t* d = op.d;
    p.B = d[S2K_WGRAD_D_B]; p.M = d[S2K_WGRAD_D_M]; p.C = d[S2K_WGRAD_D_C]; p.CTOT = d[S2K_WGRAD_D_CTOT];
    p.H = d[S2K_WGRAD_D_H]; p.W = d[S2K_WGRAD_D_W]; p.KH = d[S2K_WGRAD_D_KH]; p.KW = d[S2K_WGRAD_D_KW];
    p.S = d[S2K_WGRAD_D_STRIDE]; p.PT = d[S2K_WGRAD_D_PAD_T]; p.PL = d[S2K_WGRAD_D_PAD_L];
    p.HO = d[S2K_WGRAD_D_HO]; p.WO = d[S2K_WGRAD_D_WO]; p.prop = d[S2K_WGRAD_D_PROP]; p.proq = d[S2K_WGRAD_D_PROQ];
    const int mode = d[S2K_WGRAD_D_MODE];
    p.T = p.KH * p.KW;
    p.HWp = p.HO * p.WO;
    p.HWq = p.H * p.W;
    if (!p.p || !p.q || !p.wgs || p.B <= 0 || p.M <= 0 || p.C <= 0) { set_error("wgrad: missing tensor / bad dims"); return S2K_EINVAL; }
    if ((p.prop != S2K_PRO_NONE && !p.bnvp) || (p.proq != S2K_PRO_NONE && !p.bnvq)) {
        set_error("wgrad: prologue without BNV"); return S2K_EINVAL;
    }
    p.R = p.XW = p.XWe = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = 0;
    hipStream_t st = c.stream;
    const int64_t npix = (int64_t)p.B * p.HWp;

    if (mode == S2K_MODE_GATHER2X2) {
        if (p.T != 4 || p.H != 2 * p.HO || p.W != 2 * p.WO || p.proq != S2K_PRO_NONE || p.gateq) {
            set_error("wgrad: gather geometry"); return S2K_EINVAL;
        }
        p.NP = 32;
        p.PSTR = p.NP + 1;
        p.CSQ = 4 * p.NP + 1;
        p.ntiles = (int)cdiv64(npix, p.NP);
        if (p.M <= 32 || p.C <= 32) return launch_wg<WG_GATHER, 4, 1, 1, 2, 1, 2>(p, st);
        return launch_wg<WG_GATHER, 4, 1, 2, 2, 2, 1>(p, st);
    }
    if (p.T == 1 && p.S == 1) {
        if (p.H != p.HO || p.W != p.WO) { set_error("wgrad: 1x1 geometry"); return S2K_EINVAL; }
        p.NP = 64;
        p.PSTR = p.NP + 1;
        p.CSQ = p.NP + 1;
        p.ntiles = (int)cdiv64(npix, p.NP);
        if (p.M <= 32 && p.C <= 32) return launch_wg<WG_PIX, 1, 1, 1, 1, 1, 4>(p, st);
        if (p.M <= 64 || p.C <= 64) return launch_wg<WG_PIX, 1, 1, 1, 2, 2, 1>(p, st);
        return launch_wg<WG_PIX, 1, 2, 2, 2, 2, 1>(p, st);
    }
    if (p.T != 9) { set_error("wgrad: only 1x1, 3x3 and 2x2-transpose kernels are on this path"); return S2K_EINVAL; }
    // 3x3 (stride 1 pad 1, or the stride-2 TF-SAME stem): rectangular pixel tiles of <= 64 pixels
    const int NPX = 64;
    int XW = p.WO <= NPX ? p.WO : NPX;
    int R = NPX / ((XW + 1) & ~1);
    if (R > p.HO) R = p.HO;
    if (R < 1) R = 1;
    for (;; --R) {
        p.R = R; p.XW = XW; p.XWe = (XW + 1) & ~1;
        p.IR = (R - 1) * p.S + p.KH;
        p.IC = (p.XWe - 1) * p.S + p.KW;
        p.WS = p.IC;
        if (p.IR * p.WS <= NTHREADS * WG_EPT_MAX) break;
        if (R == 1) {
            if (XW > 16) { XW /= 2; R = 2; continue; }
            set_error("wgrad: halo tile does not fit"); return S2K_EINVAL;
        }
    }
    p.CSQ = (p.IR * p.WS) | 1;
    p.PSTR = (p.R * p.XWe) | 1;
    p.tiles_x = cdiv(p.WO, p.XW);
    p.tiles_y = cdiv(p.HO, p.R);
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.NP = 0;
    if (p.M <= 32 && p.C <= 32) return launch_wg<WG_SPATIAL, 9, 1, 1, 1, 1, 4>(p, st);
    if (p.M <= 32) return launch_wg<WG_SPATIAL, 9, 1, 1, 1, 2, 2>(p, st);
    return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 2, 1>(p, st);
}

}  // namespace s2k
