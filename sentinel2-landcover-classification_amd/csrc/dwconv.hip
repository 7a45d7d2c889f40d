// Depthwise KxK convolution (k in {3,5}, stride in {1,2}, TF-"SAME" asymmetric padding) — the
// HBM-bound part of every MBConv block (reference efficientnet_unet.py:335-352 via
// Conv2dSamePadding.forward :288-297, groups = channels).
//
// One workgroup owns PPB consecutive (b,c) planes x a band of RT output rows (whole small planes,
// several per workgroup; row bands of large ones).  The input band + halo is read ONCE from HBM,
// coalesced along W, normalised + SiLU'd on the fly (the producing conv stored raw values) and
// staged in LDS with zero padding; each wave then works on 64-output chunks of ONE plane, so the
// BatchNorm statistics of the output (forward), the BN-backward sums of the input gradient
// (dgrad) and the K*K weight-gradient sums (wgrad) are wave shuffle reductions + one f64/f32
// atomic per plane.
#include "common.h"

namespace s2k {

struct DwP {
    const float* x;      // fwd/wgrad: input;  dgrad: XRAW (raw producer output) or null
    const float* bnv;    // [4][C] of the input's BatchNorm (scale, shift, mean, invstd) or null
    const float* w;      // [C][K][K]
    const float* dy;     // dgrad/wgrad
    float* out;          // fwd: Y; dgrad: G; wgrad: DW (atomics)
    double* stats;       // fwd: [2][C] sum/sumsq of Y;  dgrad: [2][C] sum g, sum g*xhat
    int B, C, H, W, K, S, PT, PL, HO, WO, pro, beta, nrep;
    int PPB, RT, bands, IRt, ICt, LW;   // tiling
};

constexpr int DW_MAXK2 = 25;

// ---- forward ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NTHREADS) dwconv_fwd_kernel(const DwP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [PPB][IRt][LW]
    float* wsm = smem + p.PPB * p.IRt * p.LW;         // [PPB][K*K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K2 = p.K * p.K;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int yo0 = band * p.RT;
    const int rows = min(p.RT, p.HO - yo0);
    const int iy0 = yo0 * p.S - p.PT;
    const int ix0 = -p.PL;
    const int per_plane = p.IRt * p.LW;

    // stage input band (+halo) of every plane, prologue applied, zeros outside the image
    for (int idx = tid; idx < p.PPB * per_plane; idx += NTHREADS) {
        const int pl = idx / per_plane, e = idx - pl * per_plane;
        const int rr = e / p.LW, cc = e - rr * p.LW;
        const int64_t plane = pl0 + pl;
        float v = 0.0f;
        const int iy = iy0 + rr, ix = ix0 + cc;
        if (plane < nplanes && cc < p.ICt && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            const int c = (int)(plane % p.C);
            v = p.x[plane * p.H * p.W + (int64_t)iy * p.W + ix];
            if (p.pro != S2K_PRO_NONE) v = apply_pro(v, p.pro, p.bnv[c], p.bnv[p.C + c]);
        }
        tile[idx] = v;
    }
    for (int idx = tid; idx < p.PPB * K2; idx += NTHREADS) {
        const int pl = idx / K2;
        const int64_t plane = pl0 + pl;
        wsm[idx] = plane < nplanes ? p.w[(plane % p.C) * K2 + (idx - pl * K2)] : 0.0f;
    }
    __syncthreads();

    const int n_out = rows * p.WO;
    const int chunks_per_plane = (n_out + 63) >> 6;
    // waves own whole planes when there are >= 4 of them, else they split one plane's chunks; either
    // way a wave keeps its running sums in registers and issues ONE atomic pair per plane it touched
    const int wpp = p.PPB >= 4 ? 1 : 4 / p.PPB;
    double* st = p.stats ? p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.C : nullptr;
    for (int pl = wave / wpp; pl < p.PPB; pl += 4 / wpp) {
        const int64_t plane = pl0 + pl;
        if (plane >= nplanes) break;
        float s = 0.0f, q = 0.0f;
        const float* wk = wsm + pl * K2;
        for (int chn = wave % wpp; chn < chunks_per_plane; chn += wpp) {
            const int o = chn * 64 + lane;
            if (o < n_out) {
                const int r = o / p.WO, xo = o - r * p.WO;
                const float* t0 = tile + pl * per_plane + (r * p.S) * p.LW + xo * p.S;
                float acc = 0.0f;
                if (p.K == 3) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[ky * 3 + kx], t0[ky * p.LW + kx], acc);
                } else {
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) acc = fmaf(wk[ky * 5 + kx], t0[ky * p.LW + kx], acc);
                }
                p.out[plane * p.HO * p.WO + (int64_t)(yo0 + r) * p.WO + xo] = acc;
                s += acc;
                q = fmaf(acc, acc, q);
            }
        }
        if (st) {
            s = wave_sum(s);
            q = wave_sum(q);
            if (lane == 0) {
                const int c = (int)(plane % p.C);
                atomic_add_d(st + c, (double)s);
                atomic_add_d(st + p.C + c, (double)q);
            }
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(NTHREADS) dwconv_wgrad_kernel(const DwP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;  // [PPB][IRt][LW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K2 = p.K * p.K;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int yo0 = band * p.RT;
    const int rows = min(p.RT, p.HO - yo0);
    const int iy0 = yo0 * p.S - p.PT;
    const int ix0 = -p.PL;
    const int per_plane = p.IRt * p.LW;
    for (int idx = tid; idx < p.PPB * per_plane; idx += NTHREADS) {
        const int pl = idx / per_plane, e = idx - pl * per_plane;
        const int rr = e / p.LW, cc = e - rr * p.LW;
        const int64_t plane = pl0 + pl;
        float v = 0.0f;
        const int iy = iy0 + rr, ix = ix0 + cc;
        if (plane < nplanes && cc < p.ICt && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            const int c = (int)(plane % p.C);
            v = p.x[plane * p.H * p.W + (int64_t)iy * p.W + ix];
            if (p.pro != S2K_PRO_NONE) v = apply_pro(v, p.pro, p.bnv[c], p.bnv[p.C + c]);
        }
        tile[idx] = v;
    }
    __syncthreads();
    const int n_out = rows * p.WO;
    const int chunks_per_plane = (n_out + 63) >> 6;
    // a wave keeps the K*K partial sums of ONE plane in registers across that plane's chunks
    // (with fewer than 4 planes per workgroup, several waves share a plane and split its chunks)
    const int wpp = p.PPB >= 4 ? 1 : 4 / p.PPB;
    for (int pl = wave / wpp; pl < p.PPB; pl += 4 / wpp) {
        const int64_t plane = pl0 + pl;
        if (plane >= nplanes) break;
        float acc[DW_MAXK2];
#pragma unroll
        for (int i = 0; i < DW_MAXK2; ++i) acc[i] = 0.0f;
        for (int chn = wave % wpp; chn < chunks_per_plane; chn += wpp) {
            const int o = chn * 64 + lane;
            if (o < n_out) {
                const int r = o / p.WO, xo = o - r * p.WO;
                const float g = p.dy[plane * p.HO * p.WO + (int64_t)(yo0 + r) * p.WO + xo];
                const float* t0 = tile + pl * per_plane + (r * p.S) * p.LW + xo * p.S;
                if (p.K == 3) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(g, t0[ky * p.LW + kx], acc[ky * 3 + kx]);
                } else {
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] = fmaf(g, t0[ky * p.LW + kx], acc[ky * 5 + kx]);
                }
            }
        }
        const int c = (int)(plane % p.C);
#pragma unroll
        for (int i = 0; i < DW_MAXK2; ++i) {
            if (i < K2) {
                const float v = wave_sum(acc[i]);
                if (lane == 0) atomicAdd(p.out + (int64_t)c * K2 + i, v);
            }
        }
    }
}

// ---- data gradient (+ fused act' and BN-backward sums of the producer) -----------------------------
// tiles over INPUT rows: block = PPB planes x RT input rows; LDS holds the needed dY rows.
__global__ void __launch_bounds__(NTHREADS) dwconv_dgrad_kernel(const DwP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [PPB][IRt][LW]  (dY rows, zero outside)
    float* wsm = smem + p.PPB * p.IRt * p.LW;         // [PPB][K*K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K2 = p.K * p.K;
    const int64_t nplanes = (int64_t)p.B * p.C;
    const int64_t pl0 = (int64_t)(blockIdx.x / p.bands) * p.PPB;
    const int band = blockIdx.x % p.bands;
    const int iy0 = band * p.RT;                 // first input row of the band
    const int rows = min(p.RT, p.H - iy0);
    // output rows that can touch input rows [iy0, iy0+rows): yo*S + ky - PT = iy
    int yo_lo = iy0 + p.PT - (p.K - 1);
    yo_lo = yo_lo >= 0 ? (yo_lo + p.S - 1) / p.S : 0;
    const int per_plane = p.IRt * p.LW;
    for (int idx = tid; idx < p.PPB * per_plane; idx += NTHREADS) {
        const int pl = idx / per_plane, e = idx - pl * per_plane;
        const int rr = e / p.LW, cc = e - rr * p.LW;
        const int64_t plane = pl0 + pl;
        const int yo = yo_lo + rr;
        float v = 0.0f;
        if (plane < nplanes && cc < p.WO && yo < p.HO) v = p.dy[plane * p.HO * p.WO + (int64_t)yo * p.WO + cc];
        tile[idx] = v;
    }
    for (int idx = tid; idx < p.PPB * K2; idx += NTHREADS) {
        const int pl = idx / K2;
        const int64_t plane = pl0 + pl;
        wsm[idx] = plane < nplanes ? p.w[(plane % p.C) * K2 + (idx - pl * K2)] : 0.0f;
    }
    __syncthreads();
    const int n_in = rows * p.W;
    const int chunks_per_plane = (n_in + 63) >> 6;
    const int wpp = p.PPB >= 4 ? 1 : 4 / p.PPB;
    double* st = p.stats ? p.stats + (int64_t)(blockIdx.x % p.nrep) * 2 * p.C : nullptr;
    for (int pl = wave / wpp; pl < p.PPB; pl += 4 / wpp) {
        const int64_t plane = pl0 + pl;
        if (plane >= nplanes) break;
        const int c = (int)(plane % p.C);
        float s1 = 0.0f, s2 = 0.0f;
        float scale = 1.0f, shift = 0.0f, mean = 0.0f, invstd = 1.0f;
        if (p.pro != S2K_PRO_NONE) { scale = p.bnv[c]; shift = p.bnv[p.C + c]; mean = p.bnv[2 * p.C + c]; invstd = p.bnv[3 * p.C + c]; }
        const float* tp = tile + pl * per_plane;
        const float* wk = wsm + pl * K2;
        for (int chn = wave % wpp; chn < chunks_per_plane; chn += wpp) {
            const int o = chn * 64 + lane;
            if (o < n_in) {
                const int r = o / p.W, ix = o - r * p.W;
                const int iy = iy0 + r;
                float acc = 0.0f;
                for (int ky = 0; ky < p.K; ++ky) {
                    const int ty = iy + p.PT - ky;
                    if (ty < 0) continue;
                    const int yo = (p.S == 1) ? ty : (ty >> 1);
                    if ((p.S == 2 && (ty & 1)) || yo >= p.HO) continue;
                    const int rr = yo - yo_lo;   // >= 0 by construction of yo_lo
                    for (int kx = 0; kx < p.K; ++kx) {
                        const int tx = ix + p.PL - kx;
                        if (tx < 0) continue;
                        const int xo = (p.S == 1) ? tx : (tx >> 1);
                        if ((p.S == 2 && (tx & 1)) || xo >= p.WO) continue;
                        acc = fmaf(wk[ky * p.K + kx], tp[rr * p.LW + xo], acc);
                    }
                }
                const int64_t off = plane * p.H * p.W + (int64_t)iy * p.W + ix;
                if (p.pro != S2K_PRO_NONE) {
                    const float xr = p.x[off];
                    acc *= act_grad(fmaf(xr, scale, shift), p.pro);
                    s1 += acc;
                    s2 = fmaf(acc, (xr - mean) * invstd, s2);
                }
                if (p.beta) acc += p.out[off];
                p.out[off] = acc;
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
    p.HO = d[8]; p.WO = d[9]; p.pro = d[10]; p.nrep = 1;
    if (p.B <= 0 || p.C <= 0 || p.H <= 0 || p.W <= 0 || (p.K != 3 && p.K != 5) || (p.S != 1 && p.S != 2)) {
        set_error("dwconv: unsupported geometry K=%d S=%d", p.K, p.S);
        return S2K_EINVAL;
    }
    return S2K_OK;
}

// tiling over OUTPUT rows (fwd, wgrad)
static size_t tile_out(DwP& p, bool with_w) {
    const int target = 1024;
    const int hw = p.HO * p.WO;
    if (hw <= target) { p.PPB = target / hw; if (p.PPB > 16) p.PPB = 16; p.RT = p.HO; }
    else { p.PPB = 1; p.RT = target / p.WO; if (p.RT < 1) p.RT = 1; }
    p.bands = cdiv(p.HO, p.RT);
    p.IRt = (p.RT - 1) * p.S + p.K;
    p.ICt = (p.WO - 1) * p.S + p.K;
    p.LW = p.ICt | 1;
    return ((size_t)p.PPB * p.IRt * p.LW + (with_w ? p.PPB * p.K * p.K : 0)) * sizeof(float);
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
    const size_t lds = tile_out(p, true);
    if (lds > 64 * 1024) { set_error("dwconv_fwd: tile too large"); return S2K_EINVAL; }
    const int64_t groups = cdiv64((int64_t)p.B * p.C, p.PPB);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((unsigned)(groups * p.bands)), dim3(NTHREADS), lds, c.stream, p);
    return S2K_OK;
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
    const size_t lds = tile_out(p, false);
    if (lds > 64 * 1024) { set_error("dwconv_wgrad: tile too large"); return S2K_EINVAL; }
    const int64_t groups = cdiv64((int64_t)p.B * p.C, p.PPB);
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((unsigned)(groups * p.bands)), dim3(NTHREADS), lds, c.stream, p);
    return S2K_OK;
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
    // tiling over INPUT rows
    const int target = 1024;
    const int hw = p.H * p.W;
    if (hw <= target) { p.PPB = target / hw; if (p.PPB > 16) p.PPB = 16; p.RT = p.H; }
    else { p.PPB = 1; p.RT = target / p.W; if (p.RT < 1) p.RT = 1; }
    p.bands = cdiv(p.H, p.RT);
    p.IRt = (p.RT + p.K - 2) / p.S + 2;   // dY rows a band of RT input rows can touch
    if (p.IRt > p.HO) p.IRt = p.HO;
    p.ICt = p.WO;
    p.LW = p.WO | 1;
    const size_t lds = ((size_t)p.PPB * p.IRt * p.LW + p.PPB * p.K * p.K) * sizeof(float);
    if (lds > 64 * 1024) { set_error("dwconv_dgrad: tile too large"); return S2K_EINVAL; }
    const int64_t groups = cdiv64((int64_t)p.B * p.C, p.PPB);
    hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3((unsigned)(groups * p.bands)), dim3(NTHREADS), lds, c.stream, p);
    return S2K_OK;
}

}  // namespace s2k
