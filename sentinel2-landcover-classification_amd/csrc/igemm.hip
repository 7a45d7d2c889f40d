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
#include "common.h"

namespace s2k {

constexpr int KC = 8;        // input channels per K chunk
constexpr int EPT_MAX = 5;   // halo-tile elements per thread per channel (<= 1280 floats / channel)

enum { BM_PIX = 0, BM_SPATIAL = 1 };

struct ConvP {
    const float *x1, *bnv1, *gate1, *x2, *bnv2, *wt, *bias;
    float* y;
    double* stats;
    int B, C1, C2, H, W, M, KH, KW, S, PT, PL, HO, WO;
    int pro1, pro2, mode, w_sm, w_sk, w_st, flip, beta, YC;
    int T, KT, Ctot, n_mtiles, HW, Ntot, a_mfast, a_floats;
    int R, XW, tiles_x, tiles_y, IR, IC, WS, CS;
};

__device__ __forceinline__ float load_src(const ConvP& p, int c, int64_t off1, int64_t off2, int64_t cstride,
                                          int gate_row) {
    // value of concat channel c at the element whose within-plane offsets are off1/off2
    if (c < p.C1) {
        float v = p.x1[off1 + (int64_t)c * cstride];
        if (p.pro1 != S2K_PRO_NONE) v = apply_pro(v, p.pro1, p.bnv1[c], p.bnv1[p.C1 + c]);
        if (p.gate1) v *= p.gate1[gate_row + c];
        return v;
    }
    const int c2 = c - p.C1;
    float v = p.x2[off2 + (int64_t)c2 * cstride];
    if (p.pro2 != S2K_PRO_NONE) v = apply_pro(v, p.pro2, p.bnv2[c2], p.bnv2[p.C2 + c2]);
    return v;
}

template <int BMODE, int WM, int WN, int WVM, int WVN>
__global__ void __launch_bounds__(NTHREADS) conv_igemm_kernel(const ConvP p) {
    constexpr int BM = WM * WVM * 32;
    constexpr int BN = WN * WVN * 32;
    constexpr int AS = BM + 1;
    static_assert(WVM * WVN == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + p.a_floats;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WVN) * (WM * 32);
    const int wn0 = (wave % WVN) * (WN * 32);
    const int mt = blockIdx.x % p.n_mtiles;
    const int nt = blockIdx.x / p.n_mtiles;
    const int m0 = mt * BM;
    const int T = p.T, KT = p.KT;
    const bool gather = (p.mode == S2K_MODE_GATHER2X2);
    const bool scatter = (p.mode == S2K_MODE_CONVT_SCATTER);
    const int HWo = p.HO * p.WO;

    // ---- per-lane output columns (MFMA B operand columns == epilogue pixels) -----------------
    int boff[WN];
    bool cval[WN];
    int64_t ycol[WN];
    // ---- stager bookkeeping -----------------------------------------------------------------
    // PIX: one pixel per thread;  SPATIAL: up to EPT_MAX halo elements per thread
    int sp_goff[EPT_MAX];
    int64_t st_off1 = 0, st_off2 = 0, st_cstride = 0;
    int st_gate = 0;
    bool st_valid = false;
    int sb = 0;  // SPATIAL: image index of this tile

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
        const int j = tid % BN;
        const int n = n0 + j;
        st_valid = n < p.Ntot;
        const int nn = st_valid ? n : 0;
        const int b = nn / p.HW, pp = nn - b * p.HW;
        st_gate = b * p.C1;
        if (gather) {  // X1 is [B][C1/4][2H][2W]; pseudo-channel k=(co,dy,dx) handled in the chunk loop
            const int yy = pp / p.W, xx = pp - yy * p.W;
            st_off1 = (int64_t)b * (p.C1 / 4) * 4 * p.HW + (int64_t)(2 * yy) * (2 * p.W) + 2 * xx;
        } else {
            st_off1 = (int64_t)b * p.C1 * p.HW + pp;
            st_off2 = (int64_t)b * p.C2 * p.HW + pp;
        }
        st_cstride = p.HW;
    } else {
        const int tx = nt % p.tiles_x;
        const int ty = (nt / p.tiles_x) % p.tiles_y;
        sb = nt / (p.tiles_x * p.tiles_y);
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
        for (int i = 0; i < EPT_MAX; ++i) {
            const int e = tid + NTHREADS * i;
            int g = -1;
            if (e < used) {
                const int rr = e / p.WS, cc = e - rr * p.WS;
                const int iy = iy0 + rr, ix = ix0 + cc;
                if (cc < p.IC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) g = iy * p.W + ix;
            }
            sp_goff[i] = g;
        }
        st_off1 = (int64_t)sb * p.C1 * p.H * p.W;
        st_off2 = (int64_t)sb * p.C2 * p.H * p.W;
        st_cstride = (int64_t)p.H * p.W;
        st_gate = sb * p.C1;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int CSB = (BMODE == BM_PIX) ? BN : p.CS;
    const int nchunks = (p.Ctot + KC - 1) / KC;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * KC;
        __syncthreads();  // all MFMA reads of the previous chunk are done
        // ---------------- A tile: As[kk][m] = Wv[m0+m][c0 + kk/T][kk%T] ------------------------
        {
            const int total = KT * BM;
            if (p.a_mfast) {
                for (int idx = tid; idx < total; idx += NTHREADS) {
                    const int m = idx % BM, kk = idx / BM;
                    const int kc = (T == 1) ? kk : ((T == 9) ? kk / 9 : kk / T);
                    const int tap = kk - kc * T;
                    const int gm = m0 + m, c = c0 + kc;
                    float v = 0.0f;
                    if (gm < p.M && c < p.Ctot)
                        v = p.wt[(int64_t)gm * p.w_sm + (int64_t)c * p.w_sk + (p.flip ? T - 1 - tap : tap) * p.w_st];
                    As[kk * AS + m] = v;
                }
            } else {
                for (int idx = tid; idx < total; idx += NTHREADS) {
                    const int m = idx / KT, kk = idx - m * KT;
                    const int kc = (T == 1) ? kk : ((T == 9) ? kk / 9 : kk / T);
                    const int tap = kk - kc * T;
                    const int gm = m0 + m, c = c0 + kc;
                    float v = 0.0f;
                    if (gm < p.M && c < p.Ctot)
                        v = p.wt[(int64_t)gm * p.w_sm + (int64_t)c * p.w_sk + (p.flip ? T - 1 - tap : tap) * p.w_st];
                    As[kk * AS + m] = v;
                }
            }
        }
        // ---------------- B tile --------------------------------------------------------------
        if (BMODE == BM_PIX) {
            constexpr int KSTEP = NTHREADS / BN;
            const int j = tid % BN, kq = tid / BN;
#pragma unroll
            for (int i = 0; i < KC / KSTEP; ++i) {
                const int kc = kq + i * KSTEP;
                const int c = c0 + kc;
                float v = 0.0f;
                if (st_valid && c < p.Ctot) {
                    if (gather) {
                        const int co = c >> 2, dy = (c >> 1) & 1, dx = c & 1;
                        v = p.x1[st_off1 + (int64_t)co * 4 * p.HW + dy * (2 * p.W) + dx];
                    } else {
                        v = load_src(p, c, st_off1, st_off2, st_cstride, st_gate);
                    }
                }
                Bs[kc * BN + j] = v;
            }
        } else {
            const int used = p.IR * p.WS;
            for (int kc = 0; kc < KC; ++kc) {
                const int c = c0 + kc;
                const bool cok = c < p.Ctot;
#pragma unroll
                for (int i = 0; i < EPT_MAX; ++i) {
                    const int e = tid + NTHREADS * i;
                    if (e < used) {
                        float v = 0.0f;
                        if (cok && sp_goff[i] >= 0)
                            v = load_src(p, c, st_off1 + sp_goff[i], st_off2 + sp_goff[i], st_cstride, st_gate);
                        Bs[kc * p.CS + e] = v;
                    }
                }
            }
        }
        __syncthreads();
        // ---------------- MFMA ------------------------------------------------------------------
        int tdy = 0, tdx = 0;
        for (int tap = 0; tap < T; ++tap) {
            const int toff = (BMODE == BM_PIX) ? 0 : tdy * p.WS + tdx;
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) {
                const int k = 2 * ks + lh;
                float a[WM], b[WN];
#pragma unroll
                for (int rm = 0; rm < WM; ++rm) a[rm] = As[(k * T + tap) * AS + wm0 + rm * 32 + l31];
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) b[rn] = Bs[k * CSB + boff[rn] + toff];
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm], b[rn], acc[rm][rn], 0, 0, 0);
            }
            if (++tdx == p.KW) { tdx = 0; ++tdy; }
        }
    }

    // ---------------- epilogue ---------------------------------------------------------------------
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
                    if (p.beta) v += *dst;
                    *dst = v;
                    s += v;
                    q += v * v;
                }
            }
            if (p.stats) {
                s = half_sum(s);
                q = half_sum(q);
                if (l31 == 0 && rok) {
                    atomic_add_d(p.stats + gm, (double)s);
                    atomic_add_d(p.stats + p.M + gm, (double)q);
                }
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

template <int BMODE, int WM, int WN, int WVM, int WVN>
static int launch_cfg(ConvP& p, int n_ntiles, size_t b_floats, hipStream_t st) {
    constexpr int BM = WM * WVM * 32;
    p.n_mtiles = cdiv(p.M, BM);
    p.a_floats = (p.KT * (BM + 1) + 3) & ~3;
    const size_t lds = (p.a_floats + b_floats) * sizeof(float);
    auto kern = conv_igemm_kernel<BMODE, WM, WN, WVM, WVN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (lds > 160 * 1024) { set_error("conv: LDS %zu too large", lds); return S2K_EINVAL; }
    const int64_t blocks = (int64_t)p.n_mtiles * n_ntiles;
    if (blocks <= 0 || blocks > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)blocks); return S2K_EINVAL; }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHREADS), lds, st, p);
    return S2K_OK;
}

// tile geometry for the SPATIAL stager; BN = output pixels per tile
static bool spatial_tiling(ConvP& p, int BN) {
    p.XW = p.WO <= BN ? p.WO : BN;
    int R = BN / p.XW;
    if (R > p.HO) R = p.HO;
    if (R < 1) R = 1;
    for (;; --R) {
        p.R = R;
        p.IR = (R - 1) * p.S + p.KH;
        p.IC = (p.XW - 1) * p.S + p.KW;
        p.WS = p.IC;
        if (p.IR * p.WS <= NTHREADS * EPT_MAX) break;
        if (R == 1) return false;
    }
    p.CS = p.IR * p.WS;
    p.tiles_x = cdiv(p.WO, p.XW);
    p.tiles_y = cdiv(p.HO, p.R);
    return true;
}

int launch_conv(const S2kOp& op, const Ctx& c) {
    ConvP p;
    p.x1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_X1]);
    p.bnv1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_BNV1]);
    p.gate1 = ref_ptr<const float>(c, op.t[S2K_CONV_T_GATE1]);
    p.x2 = ref_ptr<const float>(c, op.t[S2K_CONV_T_X2]);
    p.bnv2 = ref_ptr<const float>(c, op.t[S2K_CONV_T_BNV2]);
    p.wt = ref_ptr<const float>(c, op.t[S2K_CONV_T_WT]);
    p.bias = ref_ptr<const float>(c, op.t[S2K_CONV_T_BIAS]);
    p.y = ref_ptr<float>(c, op.t[S2K_CONV_T_Y]);
    p.stats = ref_ptr<double>(c, op.t[S2K_CONV_T_STATS]);
    const void* ptrs[] = {p.x1, p.bnv1, p.gate1, p.x2, p.bnv2, p.wt, p.bias, p.y, p.stats};
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
    p.T = p.KH * p.KW;
    p.KT = KC * p.T;
    p.Ctot = p.C1 + p.C2;
    p.HW = p.H * p.W;
    p.a_mfast = p.w_sm < p.w_sk;
    p.R = p.XW = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = p.CS = 0;
    if (!p.x1 || !p.wt || !p.y || p.B <= 0 || p.C1 <= 0 || p.M <= 0 || p.H <= 0 || p.W <= 0) {
        set_error("conv: missing tensor or non-positive dimension");
        return S2K_EINVAL;
    }
    if (p.C2 > 0 && !p.x2) { set_error("conv: C2 > 0 without X2"); return S2K_EINVAL; }
    if ((p.pro1 != S2K_PRO_NONE && !p.bnv1) || (p.C2 > 0 && p.pro2 != S2K_PRO_NONE && !p.bnv2)) {
        set_error("conv: prologue without BNV"); return S2K_EINVAL;
    }
    const bool pix = (p.T == 1 && p.S == 1) || p.mode != S2K_MODE_CONV;
    if (p.mode != S2K_MODE_CONV) {
        if (p.T != 1 || p.S != 1 || p.C2 != 0 || p.HO != p.H || p.WO != p.W) {
            set_error("conv: scatter/gather modes are 1x1 over the low-resolution grid"); return S2K_EINVAL;
        }
        if (p.mode == S2K_MODE_CONVT_SCATTER && ((p.M & 3) || p.beta || p.stats)) {
            set_error("conv: scatter needs M = 4*Cout, beta = 0, no stats"); return S2K_EINVAL;
        }
        if (p.mode == S2K_MODE_GATHER2X2 && ((p.C1 & 3) || p.pro1 != S2K_PRO_NONE || p.gate1)) {
            set_error("conv: gather needs C1 = 4*Cout and no prologue"); return S2K_EINVAL;
        }
    }
    if (pix && (p.HO != p.H || p.WO != p.W || p.PT || p.PL)) { set_error("conv: 1x1 geometry mismatch"); return S2K_EINVAL; }
    const int64_t ntot = (int64_t)p.B * p.HW;
    if (ntot > 0x7fffffff) { set_error("conv: too many pixels"); return S2K_EINVAL; }
    p.Ntot = (int)ntot;

    const int bm = pick_bm(p.M);
    hipStream_t st = c.stream;
    if (pix) {
        // small problems (deep 8x8 maps): 64x64 tiles keep more CUs busy
        const int64_t tiles128 = (int64_t)cdiv(p.M, bm) * cdiv(p.Ntot, bm == 128 ? 128 : 256);
        if (bm >= 64 && tiles128 < 200) return launch_cfg<BM_PIX, 1, 1, 2, 2>(p, cdiv(p.Ntot, 64), KC * 64, st);
        if (bm == 128) return launch_cfg<BM_PIX, 2, 2, 2, 2>(p, cdiv(p.Ntot, 128), KC * 128, st);
        if (bm == 64) return launch_cfg<BM_PIX, 2, 2, 1, 4>(p, cdiv(p.Ntot, 256), KC * 256, st);
        return launch_cfg<BM_PIX, 1, 2, 1, 4>(p, cdiv(p.Ntot, 256), KC * 256, st);
    }
    if (p.T > 25 || p.S > 2) { set_error("conv: unsupported kernel %dx%d stride %d", p.KH, p.KW, p.S); return S2K_EINVAL; }
    const int bn = (bm == 128) ? 128 : 256;
    int64_t tiles = 0;
    auto geom = [&](int BN) -> bool {
        if (!spatial_tiling(p, BN)) return false;
        tiles = (int64_t)p.B * p.tiles_x * p.tiles_y;
        return true;
    };
    if (!geom(bn)) { set_error("conv: halo tile does not fit (W=%d)", p.W); return S2K_EINVAL; }
    if (bm >= 64 && (int64_t)cdiv(p.M, bm) * tiles < 200) {
        if (!geom(64)) { set_error("conv: halo tile does not fit"); return S2K_EINVAL; }
        return launch_cfg<BM_SPATIAL, 1, 1, 2, 2>(p, (int)tiles, (size_t)KC * p.CS, st);
    }
    if (bm == 128) return launch_cfg<BM_SPATIAL, 2, 2, 2, 2>(p, (int)tiles, (size_t)KC * p.CS, st);
    if (bm == 64) return launch_cfg<BM_SPATIAL, 2, 2, 1, 4>(p, (int)tiles, (size_t)KC * p.CS, st);
    return launch_cfg<BM_SPATIAL, 1, 2, 1, 4>(p, (int)tiles, (size_t)KC * p.CS, st);
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
