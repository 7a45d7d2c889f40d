// HBM-bound stages of the Prithvi MAE-ViT path on feature-major [B][C][L] activations:
// channel LayerNorm (token LayerNorm of timm's Block and the neck's Norm2d, prithvi_segmentation.py:11-20),
// GELU backward, random masking (prithvi.py:258-283) as a rank kernel + token gather / scatter, the patch
// im2col of PatchEmbed (:84-127), the MAE reconstruction loss (:333-350), the layout change at the API
// boundary and the Dropout2d keep-gate of the FCN head (prithvi_segmentation.py:106).
// Lanes always run along the contiguous token / pixel axis (coalesced 256-B wave rows).
#include <initializer_list>

#include "common.h"

namespace s2k {

template <typename T>
static T* ref_ptr(const Ctx& c, int64_t ref) {
    if (ref < 0) return nullptr;
    const int base = (int)(ref >> 56);
    const int64_t off = ref & ((1ll << 56) - 1);
    if (base >= c.n_bases || c.bases[base] == nullptr) return reinterpret_cast<T*>(1);
    return reinterpret_cast<T*>(static_cast<char*>(c.bases[base]) + off);
}
static bool bad(const void* q) { return q == reinterpret_cast<const void*>(1); }
#define CHECK_PTRS(name, ...)                                                                         \
    do {                                                                                              \
        const void* _ps[] = {__VA_ARGS__};                                                            \
        for (const void* _q : _ps)                                                                    \
            if (bad(_q)) { set_error(name ": tensor references a null base"); return S2K_EFAULT; }    \
    } while (0)

// ---------------- channel LayerNorm ---------------------------------------------------------------------
// One workgroup = 64 consecutive positions (lanes) of one sample x all C channels; the four waves split the
// channels, combine their per-position sums through LDS, then each normalises its own channels.
constexpr int LN_U = 8;
constexpr int LN_NW = 16;   // waves per workgroup: the channel loop of a tile is split 16 ways (latency-bound at ViT sizes)

struct LnP {
    const float *x, *gamma, *beta, *dy, *mr_in;
    float *y, *mr, *dx, *dgamma, *dbeta;
    const float* dxin;      // backward, ACCUM: DX = DXIN + ... (out of place) when given
    float* dsum;            // backward: DSUM[c] += sum of the new DX values (a bias gradient), or null
    int B, C, HW, tiles_per_b, accum;
    float eps;
    int dbg;                       // tuning build: S2K_LN_DBG bit 0 skips the all-channel sums, bit 1 the normalising pass (timing only)
    int qw, rw, pgroups, csplit;   // row kernels: 16-byte columns per position group, rows per wave, groups, channel splits
};

// ---- row kernels (HW % 4 == 0): short token rows, e.g. the MAE encoder's [768][52] samples ------------------------------
// The tile kernels above put 64 POSITIONS on the lanes of a workgroup that owns all C channels: 64 workgroups for 64 samples of
// 52 tokens, a quarter of the chip, each pulling 208-byte row pieces.  Here the lanes of a wave hold RW whole rows side by side
// (lane = row slot x 16-byte column: contiguous bytes when one position group covers the row) and a sample is shared by
// PGROUPS x CSPLIT workgroups: position groups are independent; a channel split recomputes the per-position sums over ALL
// channels (the sample's other workgroups sit on the same XCD - workgroup id mod 8 - so those reads are L2 hits) and then
// normalises only its own channel range.  No workgroup waits for another.
constexpr int LNR_NW = 16;
constexpr int LNR_U = 4;      // rows in flight per lane and tensor when normalising
constexpr int LNR_US = 8;     // ... when summing over all channels

struct LnRowsWg {
    int b, pg, cs, r, q, col;
    bool on;
};
__device__ __forceinline__ bool ln_rows_wg(const LnP& p, LnRowsWg& g) {
    const int lane = threadIdx.x & 63;
    const int group = p.pgroups * p.csplit;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int sub = k % group;
    g.b = (k / group) * 8 + xcd;
    g.pg = sub / p.csplit;
    g.cs = sub - g.pg * p.csplit;
    g.r = lane / p.qw;
    g.q = lane - g.r * p.qw;
    g.col = g.pg * p.qw + g.q;
    g.on = g.r < p.rw && g.col < (p.HW >> 2);
    return g.b < p.B;
}

__global__ void __launch_bounds__(64 * LNR_NW) chan_ln_fwd_rows_kernel(const LnP p) {
    __shared__ double red[LNR_NW][64][4];
    __shared__ double tot[64][8];
    LnRowsWg g;
    if (!ln_rows_wg(p, g)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpi = LNR_NW * p.rw;
    const int64_t base = (int64_t)g.b * p.C * p.HW + (g.on ? g.col * 4 : 0);
    const int row0 = wave * p.rw + (g.on ? g.r : 0);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
    // every channel split sums over ALL rows, starting at its own range and wrapping: the splits of a sample then pull different
    // quarters from HBM at any moment and find the rest in L2
    const int crot = (p.C + rpi * LNR_US - 1) / (rpi * LNR_US) * (rpi * LNR_US);
    const int rot = g.cs * ((p.C + p.csplit - 1) / p.csplit);
    for (int i0 = row0; i0 - row0 < p.C && !(p.dbg & 1); i0 += rpi * LNR_US) {           // uniform trip count
        float4 v[LNR_US];
        int cc[LNR_US];
#pragma unroll
        for (int u = 0; u < LNR_US; ++u) {
            int c = rot + i0 + rpi * u;
            cc[u] = c >= crot ? c - crot : c;
            v[u] = *reinterpret_cast<const float4*>(p.x + base + (int64_t)min(cc[u], p.C - 1) * p.HW);
        }
#pragma unroll
        for (int u = 0; u < LNR_US; ++u) {
            const bool ok = g.on && cc[u] < p.C;
            const float a[4] = {ok ? v[u].x : 0.0f, ok ? v[u].y : 0.0f, ok ? v[u].z : 0.0f, ok ? v[u].w : 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += a[j]; sq[j] += (double)a[j] * a[j]; }
        }
    }
    // the wave's row slots first (a shuffle tree over r), then the waves through LDS
    for (int off = 1; off < p.rw; off <<= 1) {
        const bool take = (g.r & (2 * off - 1)) == 0 && g.r + off < p.rw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double a = __shfl_down(s[j], off * p.qw), b = __shfl_down(sq[j], off * p.qw);
            s[j] += take ? a : 0.0;
            sq[j] += take ? b : 0.0;
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // sums, then sums of squares, through the same 32 KB
        if (half) __syncthreads();
        if (lane < p.qw) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave][lane][j] = half ? sq[j] : s[j];
        }
        __syncthreads();
        if ((int)threadIdx.x < p.qw * 4) {
            const int q2 = threadIdx.x >> 2, v = threadIdx.x & 3;
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < LNR_NW; ++w) a += red[w][q2][v];
            tot[q2][half * 4 + v] = a;
        }
    }
    __syncthreads();
    float mf[4], rstd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double mean = tot[g.on ? g.q : 0][j] / p.C;
        double var = tot[g.on ? g.q : 0][4 + j] / p.C - mean * mean;
        if (var < 0.0) var = 0.0;
        rstd[j] = (float)(1.0 / sqrt(var + (double)p.eps));
        mf[j] = (float)mean;
    }
    if (g.on && g.cs == 0 && wave == 0 && g.r == 0) {
        float4* mr = reinterpret_cast<float4*>(p.mr + ((int64_t)g.b * p.HW + g.col * 4) * 2);
        mr[0] = make_float4(mf[0], rstd[0], mf[1], rstd[1]);
        mr[1] = make_float4(mf[2], rstd[2], mf[3], rstd[3]);
    }
    const int rows_cs = (p.C + p.csplit - 1) / p.csplit;
    const int c_lo = g.cs * rows_cs, c_hi = min(p.C, c_lo + rows_cs);
    for (int c0 = c_lo + row0; c0 - row0 < c_hi && !(p.dbg & 2); c0 += rpi * LNR_U) {
        float4 v[LNR_U];
#pragma unroll
        for (int u = 0; u < LNR_U; ++u) v[u] = *reinterpret_cast<const float4*>(p.x + base + (int64_t)min(c0 + rpi * u, p.C - 1) * p.HW);
#pragma unroll
        for (int u = 0; u < LNR_U; ++u) {
            const int c = c0 + rpi * u;
            if (g.on && c < c_hi) {
                const float ga = p.gamma[c], be = p.beta[c];
                *reinterpret_cast<float4*>(p.y + base + (int64_t)c * p.HW) =
                    make_float4(fmaf((v[u].x - mf[0]) * rstd[0], ga, be), fmaf((v[u].y - mf[1]) * rstd[1], ga, be),
                                fmaf((v[u].z - mf[2]) * rstd[2], ga, be), fmaf((v[u].w - mf[3]) * rstd[3], ga, be));
            }
        }
    }
}

// geometry of the row kernels, or false when the tile kernels should run (rows not a multiple of 16 bytes, tiny or long rows)
static bool ln_rows_geometry(LnP& p, std::initializer_list<const void*> ptrs) {
    static const int enabled = tune_int("S2K_LN_ROWS", 1);      // 0: tile kernel only; 4: every supported shape (tests, tools/exp_ln.sh)
    // Used where the tile kernel leaves most of the chip idle - one 64-position tile per sample, few samples - and only in the FORWARD.
    // Back-to-back launches (tools/exp_ln.sh): 64 x [768][52] 17 -> 12 us; 64 x [512][200] and 64 x [768][196]: equal.  A backward
    // kernel of the same form existed in round 4 (35 -> 24 us alone, 11 of them the 3 C atomics of every workgroup, which are 64-byte
    // memory transactions) and was removed: inside the MAE step it LOST (alternating runs, tools/exp_ln_step.sh: forward only
    // 44.79 ms, neither 45.01, both 45.20) - the backward is bound by MFMA capacity, the side stream's weight gradients want the CUs
    // that 256 x 1,024 LayerNorm threads occupied, while the tile kernel's 64 workgroups leave them three quarters of the chip.
    static const int dbg = tune_int("S2K_LN_DBG", 0);
    p.dbg = dbg;
    const bool all = enabled == 4;
    if (!enabled || (p.HW & 3) || p.HW < 32 || p.HW > 1024 || p.C < 64 || (!all && (p.HW > 64 || p.B >= 128))) return false;
    for (const void* q : ptrs)
        if (reinterpret_cast<uintptr_t>(q) & 15) return false;
    const int ncol = p.HW >> 2;
    p.pgroups = cdiv(ncol, 64);
    p.qw = cdiv(ncol, p.pgroups);
    if (p.qw > 32 && p.B * p.pgroups < 256 && (ncol & 1) == 0) {     // long rows: two position groups rather than a channel split
        p.pgroups *= 2;
        p.qw = cdiv(ncol, p.pgroups);
    }
    p.rw = 64 / p.qw;
    const int rpi = LNR_NW * p.rw;
    int cs = cdiv(256, p.B * p.pgroups);
    cs = std::max(1, std::min(cs, std::min(8, p.C / (rpi * 2))));   // at least two row iterations per split
    p.csplit = cs;
    return true;
}

__global__ void __launch_bounds__(64 * LN_NW) chan_ln_fwd_kernel(const LnP p) {
    __shared__ double red[LN_NW][64][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntiles = (int64_t)p.B * p.tiles_per_b;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / p.tiles_per_b);
        const int hw = (int)(tile - (int64_t)b * p.tiles_per_b) * 64 + lane;
        const bool ok = hw < p.HW;
        const float* xb = p.x + (int64_t)b * p.C * p.HW + (ok ? hw : 0);
        double s = 0.0, q = 0.0;
        for (int c0 = wave; c0 < p.C; c0 += LN_NW * LN_U) {     // LN_U independent loads in flight per lane
            float v[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) v[u] = xb[(int64_t)min(c0 + LN_NW * u, p.C - 1) * p.HW];   // unconditional (clamped) loads:
            // hipcc turns `cond ? load : 0` into a branch per load and waits for each one in turn
#pragma unroll
            for (int u = 0; u < LN_U; ++u) v[u] = (ok && c0 + LN_NW * u < p.C) ? v[u] : 0.0f;
#pragma unroll
            for (int u = 0; u < LN_U; ++u) { s += v[u]; q += (double)v[u] * v[u]; }
        }
        red[wave][lane][0] = s;
        red[wave][lane][1] = q;
        __syncthreads();
        s = 0.0;
        q = 0.0;
#pragma unroll
        for (int w = 0; w < LN_NW; ++w) { s += red[w][lane][0]; q += red[w][lane][1]; }
        const double mean = s / p.C;
        double var = q / p.C - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float mf = (float)mean;
        if (ok && wave == 0) {
            float* mr = p.mr + ((int64_t)b * p.HW + hw) * 2;
            mr[0] = mf;
            mr[1] = rstd;
        }
        float* yb = p.y + (int64_t)b * p.C * p.HW + (ok ? hw : 0);
        for (int c0 = wave; c0 < p.C; c0 += LN_NW * LN_U) {
            float v[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) v[u] = xb[(int64_t)min(c0 + LN_NW * u, p.C - 1) * p.HW];   // unconditional (clamped) loads:
            // hipcc turns `cond ? load : 0` into a branch per load and waits for each one in turn
#pragma unroll
            for (int u = 0; u < LN_U; ++u) v[u] = (ok && c0 + LN_NW * u < p.C) ? v[u] : 0.0f;
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int c = c0 + LN_NW * u;
                if (ok && c < p.C) yb[(int64_t)c * p.HW] = fmaf((v[u] - mf) * rstd, p.gamma[c], p.beta[c]);
            }
        }
        __syncthreads();
    }
}

int launch_chan_ln_fwd(const S2kOp& op, const Ctx& c) {
    LnP p{};
    p.x = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_FWD_T_X]);
    p.gamma = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_FWD_T_GAMMA]);
    p.beta = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_FWD_T_BETA]);
    p.y = ref_ptr<float>(c, op.t[S2K_CHAN_LN_FWD_T_Y]);
    p.mr = ref_ptr<float>(c, op.t[S2K_CHAN_LN_FWD_T_MR]);
    CHECK_PTRS("chan_ln_fwd", p.x, p.gamma, p.beta, p.y, p.mr);
    p.B = op.d[S2K_CHAN_LN_FWD_D_B]; p.C = op.d[S2K_CHAN_LN_FWD_D_C]; p.HW = op.d[S2K_CHAN_LN_FWD_D_HW];
    p.eps = op.f[S2K_CHAN_LN_FWD_F_EPS];
    if (!p.x || !p.gamma || !p.beta || !p.y || !p.mr || p.B <= 0 || p.C <= 0 || p.HW <= 0) { set_error("chan_ln_fwd: bad args"); return S2K_EINVAL; }
    if (ln_rows_geometry(p, {p.x, p.y, p.mr})) {
        const unsigned grid = 8u * (unsigned)(p.pgroups * p.csplit) * (unsigned)cdiv(p.B, 8);
        hipLaunchKernelGGL(chan_ln_fwd_rows_kernel, dim3(grid), dim3(64 * LNR_NW), 0, c.stream, p);
        return S2K_OK;
    }
    p.tiles_per_b = cdiv(p.HW, 64);
    const int64_t tiles = (int64_t)p.B * p.tiles_per_b;
    hipLaunchKernelGGL(chan_ln_fwd_kernel, dim3((unsigned)std::min<int64_t>(tiles, 8192)), dim3(64 * LN_NW), 0, c.stream, p);
    return S2K_OK;
}

// backward: dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * gamma; the per-channel parameter
// sums of a workgroup's tiles are collected in LDS (each wave owns its channels) and flushed once at the end
__global__ void __launch_bounds__(64 * LN_NW) chan_ln_bwd_kernel(const LnP p) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float* pg = lsm;              // [C] dgamma partial
    float* pb = lsm + p.C;        // [C] dbeta partial
    float* pd = lsm + 2 * p.C;    // [C] sum of the new DX values (DSUM)
    float(*red)[64][2] = reinterpret_cast<float(*)[64][2]>(lsm + 3 * p.C);   // [LN_NW][64][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool want_p = p.dgamma != nullptr, want_d = p.dsum != nullptr;
    const float* acc_src = p.dxin ? p.dxin : p.dx;
    for (int c = threadIdx.x; c < 3 * p.C; c += 64 * LN_NW) lsm[c] = 0.0f;
    __syncthreads();
    const int64_t ntiles = (int64_t)p.B * p.tiles_per_b;
    const float invC = 1.0f / p.C;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / p.tiles_per_b);
        const int hw = (int)(tile - (int64_t)b * p.tiles_per_b) * 64 + lane;
        const bool ok = hw < p.HW;
        const int64_t base = (int64_t)b * p.C * p.HW + (ok ? hw : 0);
        float mean = 0.0f, rstd = 0.0f;
        if (ok) {
            const float* mr = p.mr_in + ((int64_t)b * p.HW + hw) * 2;
            mean = mr[0];
            rstd = mr[1];
        }
        float s1 = 0.0f, s2 = 0.0f;
        for (int c0 = wave; c0 < p.C; c0 += LN_NW * LN_U) {
            float dv[LN_U], xv[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int64_t off = base + (int64_t)min(c0 + LN_NW * u, p.C - 1) * p.HW;   // clamped: loads never sit under a condition
                dv[u] = p.dy[off];
                xv[u] = p.x[off];
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const bool on = ok && c0 + LN_NW * u < p.C;
                dv[u] = on ? dv[u] : 0.0f;
                xv[u] = on ? xv[u] : mean;
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int c = c0 + LN_NW * u;
                const float g = dv[u] * (c < p.C ? p.gamma[c] : 0.0f);
                s1 += g;
                s2 = fmaf(g, (xv[u] - mean) * rstd, s2);
            }
        }
        red[wave][lane][0] = s1;
        red[wave][lane][1] = s2;
        __syncthreads();
        s1 = 0.0f;
        s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < LN_NW; ++w) { s1 += red[w][lane][0]; s2 += red[w][lane][1]; }
        const float m1 = s1 * invC, m2 = s2 * invC;
        for (int c0 = wave; c0 < p.C; c0 += LN_NW * LN_U) {
            float dv[LN_U], xv[LN_U], ov[LN_U];
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int64_t off = base + (int64_t)min(c0 + LN_NW * u, p.C - 1) * p.HW;
                dv[u] = p.dy[off];
                xv[u] = p.x[off];
                ov[u] = p.accum ? acc_src[off] : 0.0f;   // kernel-uniform condition
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const bool on = ok && c0 + LN_NW * u < p.C;
                dv[u] = on ? dv[u] : 0.0f;
                xv[u] = on ? xv[u] : mean;
            }
#pragma unroll
            for (int u = 0; u < LN_U; ++u) {
                const int c = c0 + LN_NW * u;
                if (c < p.C) {      // wave-uniform
                    const float xh = (xv[u] - mean) * rstd;
                    const float g = dv[u] * p.gamma[c];
                    const float nv = rstd * (g - m1 - xh * m2) + ov[u];
                    if (ok) p.dx[base + (int64_t)c * p.HW] = nv;
                    if (want_d) {
                        const float sd = wave_sum_hi(ok ? nv : 0.0f);
                        if (lane == 63) pd[c] += sd;
                    }
                    if (want_p) {
                        const float a = wave_sum_hi(dv[u] * xh), bb = wave_sum_hi(dv[u]);
                        if (lane == 63) { pg[c] += a; pb[c] += bb; }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (want_p || want_d) {
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += 64 * LN_NW) {
            if (want_p) {
                atomicAdd(p.dgamma + c, pg[c]);
                atomicAdd(p.dbeta + c, pb[c]);
            }
            if (want_d) atomicAdd(p.dsum + c, pd[c]);
        }
    }
}

int launch_chan_ln_bwd(const S2kOp& op, const Ctx& c) {
    LnP p{};
    p.dy = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_BWD_T_DY]);
    p.x = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_BWD_T_X]);
    p.mr_in = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_BWD_T_MR]);
    p.gamma = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_BWD_T_GAMMA]);
    p.dx = ref_ptr<float>(c, op.t[S2K_CHAN_LN_BWD_T_DX]);
    p.dgamma = ref_ptr<float>(c, op.t[S2K_CHAN_LN_BWD_T_DGAMMA]);
    p.dbeta = ref_ptr<float>(c, op.t[S2K_CHAN_LN_BWD_T_DBETA]);
    p.dxin = ref_ptr<const float>(c, op.t[S2K_CHAN_LN_BWD_T_DXIN]);
    p.dsum = ref_ptr<float>(c, op.t[S2K_CHAN_LN_BWD_T_DSUM]);
    CHECK_PTRS("chan_ln_bwd", p.dy, p.x, p.mr_in, p.gamma, p.dx, p.dgamma, p.dbeta, p.dxin, p.dsum);
    p.B = op.d[S2K_CHAN_LN_BWD_D_B]; p.C = op.d[S2K_CHAN_LN_BWD_D_C]; p.HW = op.d[S2K_CHAN_LN_BWD_D_HW];
    p.accum = op.d[S2K_CHAN_LN_BWD_D_ACCUM];
    if (!p.dy || !p.x || !p.mr_in || !p.gamma || !p.dx || p.B <= 0 || p.C <= 0 || p.HW <= 0 || (!p.dgamma != !p.dbeta)) {
        set_error("chan_ln_bwd: bad args"); return S2K_EINVAL;
    }
    p.tiles_per_b = cdiv(p.HW, 64);
    const int64_t tiles = (int64_t)p.B * p.tiles_per_b;
    const size_t lds = (3 * (size_t)p.C + LN_NW * 64 * 2) * sizeof(float);
    if (lds > 64 * 1024) { set_error("chan_ln_bwd: C too large"); return S2K_EINVAL; }
    hipLaunchKernelGGL(chan_ln_bwd_kernel, dim3((unsigned)std::min<int64_t>(tiles, 1024)), dim3(64 * LN_NW), lds, c.stream, p);
    return S2K_OK;
}

// ---------------- G *= act'(X) ---------------------------------------------------------------------------
__global__ void act_bwd_kernel(float* g, const float* x, int64_t n, int act) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= (act == 5 ? x[i] : act_grad(x[i], act));
}

int launch_act_bwd(const S2kOp& op, const Ctx& c) {
    float* g = ref_ptr<float>(c, op.t[S2K_ACT_BWD_T_G]);
    const float* x = ref_ptr<const float>(c, op.t[S2K_ACT_BWD_T_X]);
    CHECK_PTRS("act_bwd", g, x);
    const int64_t n = op.n[S2K_ACT_BWD_N_COUNT];
    const int act = op.d[S2K_ACT_BWD_D_ACT];
    if (!g || !x || n <= 0 || (act != S2K_PRO_GELU && act != S2K_PRO_SILU && act != S2K_PRO_RELU && act != 5)) { set_error("act_bwd: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 8192)), dim3(256), 0, c.stream, g, x, n, act);
    return S2K_OK;
}

// ---------------- Y = act(X) ------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* x, float* y, int64_t n, int act, const float* bnv, int C, int HW) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float sc = 1.0f, sh = 0.0f;
        if (bnv) {
            const int c = (int)((i / HW) % C);
            sc = bnv[c];
            sh = bnv[C + c];
        }
        y[i] = apply_pro(x[i], act, sc, sh);
    }
}

// BatchNorm + SiLU + SE gate of a whole [B][C][HW] tensor (HW % 4 == 0): one wave per plane chunk, 16-byte loads / stores, the
// plane's three constants once per wave
__global__ void __launch_bounds__(256) act_gate_planes_kernel(const float* x, float* y, const float* bnv, const float* gate, int C, int HW,
                                                              int64_t nplanes, int chunks) {
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= nplanes * chunks) return;
    const int lane = threadIdx.x & 63;
    const int64_t plane = task / chunks;
    const int ck = (int)(task - plane * chunks);
    const int c = (int)(plane % C);
    const float sc = bnv[c], sh = bnv[C + c], g = gate[plane];
    const int start = ck * 4096, n4 = (min(4096, HW - start)) >> 2;
    const float4* src = reinterpret_cast<const float4*>(x + plane * HW + start);
    float4* dst = reinterpret_cast<float4*>(y + plane * HW + start);
    for (int i = lane; i < n4; i += 64) {
        const float4 v = src[i];
        dst[i] = make_float4(silu_f(fmaf(v.x, sc, sh)) * g, silu_f(fmaf(v.y, sc, sh)) * g, silu_f(fmaf(v.z, sc, sh)) * g, silu_f(fmaf(v.w, sc, sh)) * g);
    }
}

int launch_act_fwd(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_ACT_FWD_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_ACT_FWD_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_ACT_FWD_T_BNV]);
    const float* gate = ref_ptr<const float>(c, op.t[S2K_ACT_FWD_T_GATE]);
    CHECK_PTRS("act_fwd", x, y, bnv, gate);
    if (gate) {
        const int64_t n = op.n[S2K_ACT_FWD_N_COUNT];
        const int C = op.d[S2K_ACT_FWD_D_C], HW = op.d[S2K_ACT_FWD_D_HW];
        if (!x || !y || !bnv || op.d[S2K_ACT_FWD_D_ACT] != S2K_PRO_SILU || C <= 0 || HW <= 0 || (HW & 3) || n % ((int64_t)C * HW)) {
            set_error("act_fwd: the gated form is BatchNorm + SiLU on [B][C][HW] with HW % 4 == 0"); return S2K_EINVAL;
        }
        const int64_t nplanes = n / HW;
        const int chunks = (HW + 4095) / 4096;
        if (nplanes * chunks >= 0x7fffffffll) { set_error("act_fwd: too many planes"); return S2K_EINVAL; }
        hipLaunchKernelGGL(act_gate_planes_kernel, dim3((unsigned)cdiv64(nplanes * chunks, 4)), dim3(256), 0, c.stream, x, y, bnv, gate, C, HW, nplanes, chunks);
        return S2K_OK;
    }
    const int64_t n = op.n[S2K_ACT_FWD_N_COUNT];
    const int act = op.d[S2K_ACT_FWD_D_ACT], C = op.d[S2K_ACT_FWD_D_C], HW = op.d[S2K_ACT_FWD_D_HW];
    if (!x || !y || n <= 0 || (act != S2K_PRO_GELU && act != S2K_PRO_SILU && act != S2K_PRO_RELU) || (bnv && (C <= 0 || HW <= 0))) {
        set_error("act_fwd: bad args"); return S2K_EINVAL;
    }
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 8192)), dim3(256), 0, c.stream, x, y, n, act, bnv, C, HW);
    return S2K_OK;
}

// ---------------- Dropout2d keep gate --------------------------------------------------------------------
__global__ void drop_gate_kernel(const float* u, float* gate, int64_t n, float prob) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gate[i] = u[i] >= prob ? 1.0f / (1.0f - prob) : 0.0f;
}

int launch_drop_gate(const S2kOp& op, const Ctx& c) {
    const float* u = ref_ptr<const float>(c, op.t[S2K_DROP_GATE_T_U]);
    float* gate = ref_ptr<float>(c, op.t[S2K_DROP_GATE_T_GATE]);
    CHECK_PTRS("drop_gate", u, gate);
    const int64_t n = op.n[S2K_DROP_GATE_N_COUNT];
    const float prob = op.f[S2K_DROP_GATE_F_P];
    if (!u || !gate || n <= 0 || !(prob >= 0.0f && prob < 1.0f)) { set_error("drop_gate: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(drop_gate_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, c.stream, u, gate, n, prob);
    return S2K_OK;
}

// ---------------- rank of the masking noise -> ids_restore, mask, gather tables ------------------------------
// one workgroup per sample; rank[i] = #{j : n[j] < n[i] or (n[j] == n[i] and j < i)}  (= a stable argsort)
__global__ void __launch_bounds__(NTHREADS) mae_mask_index_kernel(const float* noise, int64_t* ids_restore, float* mask, int* enc_idx,
                                                                    int* dec_idx, int L, int keep) {
    extern __shared__ __attribute__((aligned(16))) float ns[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += NTHREADS) ns[i] = noise[(int64_t)b * L + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        enc_idx[(int64_t)b * (1 + keep)] = -1;
        dec_idx[(int64_t)b * (1 + L)] = 0;
    }
    for (int i = threadIdx.x; i < L; i += NTHREADS) {
        const float v = ns[i];
        int r = 0;
        for (int j = 0; j < L; ++j) {
            const float w = ns[j];
            r += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        ids_restore[(int64_t)b * L + i] = r;
        mask[(int64_t)b * L + i] = r >= keep ? 1.0f : 0.0f;
        if (r < keep) enc_idx[(int64_t)b * (1 + keep) + 1 + r] = i;
        dec_idx[(int64_t)b * (1 + L) + 1 + i] = r < keep ? 1 + r : -1;
    }
}

int launch_mae_mask_index(const S2kOp& op, const Ctx& c) {
    const float* noise = ref_ptr<const float>(c, op.t[S2K_MAE_MASK_INDEX_T_NOISE]);
    int64_t* ids = ref_ptr<int64_t>(c, op.t[S2K_MAE_MASK_INDEX_T_IDS_RESTORE]);
    float* mask = ref_ptr<float>(c, op.t[S2K_MAE_MASK_INDEX_T_MASK]);
    int* enc = ref_ptr<int>(c, op.t[S2K_MAE_MASK_INDEX_T_ENC_IDX]);
    int* dec = ref_ptr<int>(c, op.t[S2K_MAE_MASK_INDEX_T_DEC_IDX]);
    CHECK_PTRS("mae_mask_index", noise, ids, mask, enc, dec);
    const int B = op.d[S2K_MAE_MASK_INDEX_D_B], L = op.d[S2K_MAE_MASK_INDEX_D_L], keep = op.d[S2K_MAE_MASK_INDEX_D_KEEP];
    if (!noise || !ids || !mask || !enc || !dec || B <= 0 || L <= 0 || keep < 0 || keep > L || L > 12288) { set_error("mae_mask_index: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(mae_mask_index_kernel, dim3(B), dim3(NTHREADS), (size_t)L * sizeof(float), c.stream, noise, ids, mask, enc, dec, L, keep);
    return S2K_OK;
}

// ---------------- token gather / scatter --------------------------------------------------------------------
struct TokP {
    const float *in, *fill, *pos, *dout;
    const int* idx;
    float *out, *din, *dfill;
    int B, C, Lin, Lout, pos_by_src, pos_off;
    int LinS, LoutS;   // row strides (>= Lin / Lout); columns Lout..LoutS-1 of OUT and Lin..LinS-1 of DIN are written as zeros
};

__global__ void token_gather_kernel(const TokP p) {
    const int64_t n = (int64_t)p.B * p.C * p.LoutS;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int j = (int)(i % p.LoutS);
        const int64_t bc = i / p.LoutS;
        float v = 0.0f;
        if (j < p.Lout) {
            const int cc = (int)(bc % p.C), b = (int)(bc / p.C);
            const int src = p.idx[(int64_t)b * p.Lout + j];
            v = src >= 0 ? p.in[bc * p.LinS + src] : (p.fill ? p.fill[cc] : 0.0f);
            if (p.pos) v += p.pos[(int64_t)(p.pos_by_src ? src + p.pos_off : j) * p.C + cc];
        }
        p.out[i] = v;
    }
}

int launch_token_gather(const S2kOp& op, const Ctx& c) {
    TokP p{};
    p.in = ref_ptr<const float>(c, op.t[S2K_TOKEN_GATHER_T_IN]);
    p.idx = ref_ptr<const int>(c, op.t[S2K_TOKEN_GATHER_T_IDX]);
    p.fill = ref_ptr<const float>(c, op.t[S2K_TOKEN_GATHER_T_FILL]);
    p.pos = ref_ptr<const float>(c, op.t[S2K_TOKEN_GATHER_T_POS]);
    p.out = ref_ptr<float>(c, op.t[S2K_TOKEN_GATHER_T_OUT]);
    CHECK_PTRS("token_gather", p.in, p.idx, p.fill, p.pos, p.out);
    p.B = op.d[S2K_TOKEN_GATHER_D_B]; p.C = op.d[S2K_TOKEN_GATHER_D_C]; p.Lin = op.d[S2K_TOKEN_GATHER_D_LIN];
    p.Lout = op.d[S2K_TOKEN_GATHER_D_LOUT]; p.pos_by_src = op.d[S2K_TOKEN_GATHER_D_POS_BY_SRC]; p.pos_off = op.d[S2K_TOKEN_GATHER_D_POS_OFF];
    p.LinS = op.d[S2K_TOKEN_GATHER_D_LIN_S] > 0 ? op.d[S2K_TOKEN_GATHER_D_LIN_S] : p.Lin;
    p.LoutS = op.d[S2K_TOKEN_GATHER_D_LOUT_S] > 0 ? op.d[S2K_TOKEN_GATHER_D_LOUT_S] : p.Lout;
    if (!p.in || !p.idx || !p.out || p.B <= 0 || p.C <= 0 || p.Lin <= 0 || p.Lout <= 0 || p.LinS < p.Lin || p.LoutS < p.Lout) {
        set_error("token_gather: bad args"); return S2K_EINVAL;
    }
    const int64_t n = (int64_t)p.B * p.C * p.LoutS;
    hipLaunchKernelGGL(token_gather_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 8192)), dim3(256), 0, c.stream, p);
    return S2K_OK;
}

// one wave per (b, c) row: the row of DIN is assembled in LDS (zero, scatter, copy out), the gradient of the fill
// token is a wave sum + one atomic per row
__global__ void __launch_bounds__(NTHREADS) token_scatter_kernel(const TokP p) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [4][LinS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* row = rows + (size_t)wave * p.LinS;
    const int64_t nrows = (int64_t)p.B * p.C;
    for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < nrows; r0 += (int64_t)gridDim.x * 4) {
        const int64_t r = r0 + wave;
        const bool ok = r < nrows;
        const int b = ok ? (int)(r / p.C) : 0, cc = ok ? (int)(r % p.C) : 0;
        for (int i = lane; i < p.LinS; i += 64) row[i] = 0.0f;
        __syncthreads();
        float acc = 0.0f;
        if (ok)
            for (int j = lane; j < p.Lout; j += 64) {
                const int dst = p.idx[(int64_t)b * p.Lout + j];
                const float v = p.dout[r * p.LoutS + j];
                if (dst >= 0) row[dst] = v;
                else acc += v;
            }
        __syncthreads();
        if (ok)
            for (int i = lane; i < p.LinS; i += 64) p.din[r * p.LinS + i] = row[i];
        if (p.dfill) {
            acc = wave_sum_hi(acc);
            if (ok && lane == 63) atomicAdd(p.dfill + cc, acc);
        }
        __syncthreads();
    }
}

int launch_token_scatter(const S2kOp& op, const Ctx& c) {
    TokP p{};
    p.dout = ref_ptr<const float>(c, op.t[S2K_TOKEN_SCATTER_T_DOUT]);
    p.idx = ref_ptr<const int>(c, op.t[S2K_TOKEN_SCATTER_T_IDX]);
    p.din = ref_ptr<float>(c, op.t[S2K_TOKEN_SCATTER_T_DIN]);
    p.dfill = ref_ptr<float>(c, op.t[S2K_TOKEN_SCATTER_T_DFILL]);
    CHECK_PTRS("token_scatter", p.dout, p.idx, p.din, p.dfill);
    p.B = op.d[S2K_TOKEN_SCATTER_D_B]; p.C = op.d[S2K_TOKEN_SCATTER_D_C]; p.Lin = op.d[S2K_TOKEN_SCATTER_D_LIN]; p.Lout = op.d[S2K_TOKEN_SCATTER_D_LOUT];
    p.LinS = op.d[S2K_TOKEN_SCATTER_D_LIN_S] > 0 ? op.d[S2K_TOKEN_SCATTER_D_LIN_S] : p.Lin;
    p.LoutS = op.d[S2K_TOKEN_SCATTER_D_LOUT_S] > 0 ? op.d[S2K_TOKEN_SCATTER_D_LOUT_S] : p.Lout;
    if (!p.dout || !p.idx || !p.din || p.B <= 0 || p.C <= 0 || p.Lin <= 0 || p.Lout <= 0 || p.LinS > 4096 || p.LinS < p.Lin || p.LoutS < p.Lout) {
        set_error("token_scatter: bad args"); return S2K_EINVAL;
    }
    const int64_t nrows = (int64_t)p.B * p.C;
    hipLaunchKernelGGL(token_scatter_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(nrows, 4), 4096)), dim3(NTHREADS), (size_t)4 * p.LinS * sizeof(float), c.stream, p);
    return S2K_OK;
}

// ---------------- patch geometry shared by PATCHIFY and the MAE loss -----------------------------------------
struct PatchP {
    const float *x, *pred, *mask, *gout;
    float *out, *loss, *dpred;
    double* acc;
    int B, C, T, H, W, P, TUB, LP, L_OFF, norm_pix;
    int gt, gh, gw, L, PD;
};

static int fill_patch(PatchP& p, const int32_t* d) {
    p.B = d[0]; p.C = d[1]; p.T = d[2]; p.H = d[3]; p.W = d[4]; p.P = d[5]; p.TUB = d[6];
    if (p.B <= 0 || p.C <= 0 || p.T <= 0 || p.H <= 0 || p.W <= 0 || p.P <= 0 || p.TUB <= 0 || p.T % p.TUB || p.H % p.P || p.W % p.P) {
        set_error("patch geometry: bad dims"); return S2K_EINVAL;
    }
    p.gt = p.T / p.TUB; p.gh = p.H / p.P; p.gw = p.W / p.P;
    p.L = p.gt * p.gh * p.gw;
    p.PD = p.TUB * p.P * p.P * p.C;
    return S2K_OK;
}

// OUT[b][((c*TUB + tt)*P + py)*P + px][l] ; lanes along l.  inverse: 0 OUT = f(X), 1 X = OUT, 2 X = -OUT, 3 X += OUT;
// order 1: rows in the MAE target's order (tt, py, px, c); OUT rows have stride ls, patch l sits in column l + l_off
__global__ void patchify_kernel(const PatchP p, int inverse, int order, int ls, int l_off) {
    const int64_t n = (int64_t)p.B * p.PD * p.L;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float* xw = const_cast<float*>(p.x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int l = (int)(i % p.L);
        const int64_t bk = i / p.L;
        const int k = (int)(bk % p.PD), b = (int)(bk / p.PD);
        int px, py, tt, cc;
        if (order == 0) { px = k % p.P; py = (k / p.P) % p.P; tt = (k / (p.P * p.P)) % p.TUB; cc = k / (p.P * p.P * p.TUB); }
        else { cc = k % p.C; px = (k / p.C) % p.P; py = (k / (p.C * p.P)) % p.P; tt = k / (p.C * p.P * p.P); }
        const int w = l % p.gw, h = (l / p.gw) % p.gh, t = l / (p.gw * p.gh);
        const int64_t xi = ((((int64_t)b * p.C + cc) * p.T + t * p.TUB + tt) * p.H + h * p.P + py) * p.W + w * p.P + px;
        const int64_t oi = ((int64_t)b * p.PD + k) * ls + l + l_off;
        if (inverse == 0) p.out[oi] = p.x[xi];
        else if (inverse == 1) xw[xi] = p.out[oi];
        else if (inverse == 2) xw[xi] = -p.out[oi];
        else xw[xi] += p.out[oi];
    }
}

// PATCHIFY INVERSE 4: the gradient w.r.t. the images through the standardised loss target.  One wave per patch, lanes along the
// PD features (order (tt, py, px, c)): mean, centred sum of squares (two passes, as torch.var), then the two sums over g = -OUT,
// then the PD gradient values.  Every pixel belongs to exactly one patch, so DX is written, not accumulated.
__global__ void __launch_bounds__(NTHREADS) patch_norm_target_grad_kernel(const PatchP p, const float* imgs, float* dx, const float* gpred,
                                                                         int ls, int l_off) {
    const int lane = threadIdx.x & 63;
    const int64_t patch = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (patch >= (int64_t)p.B * p.L) return;
    const int b = (int)(patch / p.L), l = (int)(patch % p.L);
    const int w = l % p.gw, h = (l / p.gw) % p.gh, t = l / (p.gw * p.gh);
    const int64_t base = (((int64_t)b * p.C * p.T + (int64_t)t * p.TUB) * p.H + (int64_t)h * p.P) * p.W + (int64_t)w * p.P;
    const int64_t cstride = (int64_t)p.T * p.H * p.W;
    auto pix = [&](int f) -> int64_t {
        const int cc = f % p.C, px = (f / p.C) % p.P, py = (f / (p.C * p.P)) % p.P, tt = f / (p.C * p.P * p.P);
        return base + (int64_t)cc * cstride + ((int64_t)tt * p.H + py) * p.W + px;
    };
    const float* g = gpred + (int64_t)b * p.PD * ls + l + l_off;
    double s = 0.0;
    for (int f = lane; f < p.PD; f += 64) s += (double)imgs[pix(f)];
    const double mu = wave_sum_d(s) / p.PD;
    double q = 0.0, sg = 0.0, sgx = 0.0;
    for (int f = lane; f < p.PD; f += 64) {
        const double xc = (double)imgs[pix(f)] - mu, gv = -(double)g[(int64_t)f * ls];
        q += xc * xc;
        sg += gv;
        sgx += gv * xc;
    }
    q = wave_sum_d(q);
    sg = wave_sum_d(sg);
    sgx = wave_sum_d(sgx);
    const double sd = sqrt(q / (p.PD - 1) + 1.0e-6);
    const double inv = 1.0 / sd, gm = sg / p.PD, k = sgx / (sd * sd * sd * (p.PD - 1));
    for (int f = lane; f < p.PD; f += 64) {
        const int64_t xi = pix(f);
        const double xc = (double)imgs[xi] - mu, gv = -(double)g[(int64_t)f * ls];
        dx[xi] = (float)((gv - gm) * inv - xc * k);
    }
}

int launch_patchify(const S2kOp& op, const Ctx& c) {
    PatchP p{};
    if (int e = fill_patch(p, op.d)) return e;
    p.x = ref_ptr<const float>(c, op.t[S2K_PATCHIFY_T_X]);
    p.out = ref_ptr<float>(c, op.t[S2K_PATCHIFY_T_OUT]);
    const float* imgs = ref_ptr<const float>(c, op.t[S2K_PATCHIFY_T_IMGS]);
    CHECK_PTRS("patchify", p.x, p.out, imgs);
    if (!p.x || !p.out) { set_error("patchify: missing tensor"); return S2K_EINVAL; }
    const int inverse = op.d[S2K_PATCHIFY_D_INVERSE], order = op.d[S2K_PATCHIFY_D_ORDER], l_off = op.d[S2K_PATCHIFY_D_L_OFF];
    const int ls = op.d[S2K_PATCHIFY_D_LS] > 0 ? op.d[S2K_PATCHIFY_D_LS] : p.L;
    if (inverse < 0 || inverse > 4 || order < 0 || order > 1 || l_off < 0 || ls < p.L + l_off) { set_error("patchify: bad mode / stride"); return S2K_EINVAL; }
    if (inverse == 4) {
        if (!imgs || order != 1 || p.PD < 2) { set_error("patchify: the standardised-target gradient needs IMGS, ORDER 1 and PD >= 2"); return S2K_EINVAL; }
        const int64_t npatch = (int64_t)p.B * p.L;
        hipLaunchKernelGGL(patch_norm_target_grad_kernel, dim3((unsigned)cdiv64(npatch, 4)), dim3(NTHREADS), 0, c.stream, p, imgs,
                           const_cast<float*>(p.x), p.out, ls, l_off);
        return S2K_OK;
    }
    const int64_t n = (int64_t)p.B * p.PD * p.L;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 16384)), dim3(256), 0, c.stream, p, inverse, order, ls, l_off);
    return S2K_OK;
}

// target value f = ((tt*P + py)*P + px)*C + c of patch l of sample b (prithvi.py:236-245)
__device__ __forceinline__ float patch_target(const PatchP& p, int b, int l, int f) {
    const int cc = f % p.C, px = (f / p.C) % p.P, py = (f / (p.C * p.P)) % p.P, tt = f / (p.C * p.P * p.P);
    const int w = l % p.gw, h = (l / p.gw) % p.gh, t = l / (p.gw * p.gh);
    return p.x[((((int64_t)b * p.C + cc) * p.T + t * p.TUB + tt) * p.H + h * p.P + py) * p.W + w * p.P + px];
}

// the same values in order f = 0, 1, 2, ... (c fastest, then px, py, tt) with incremental addressing: the loss kernels
// spent most of their time in the five divisions of patch_target
struct PatchWalk {
    const float* q;      // element (tt, py, px, c = 0)
    int cc, px, py;
    int64_t cstride, row_skip, frame_skip;
    __device__ __forceinline__ PatchWalk(const PatchP& p, int b, int l) {
        const int w = l % p.gw, h = (l / p.gw) % p.gh, t = l / (p.gw * p.gh);
        cstride = (int64_t)p.T * p.H * p.W;
        q = p.x + (((int64_t)b * p.C * p.T + (int64_t)t * p.TUB) * p.H + (int64_t)h * p.P) * p.W + (int64_t)w * p.P;
        cc = px = py = 0;
        row_skip = p.W - p.P;                                   // from the end of a patch row to the start of the next
        frame_skip = (int64_t)p.H * p.W - (int64_t)p.P * p.W;   // from below the patch to its first row in the next frame
    }
    __device__ __forceinline__ float next(const PatchP& p) {
        const float v = q[(int64_t)cc * cstride];
        if (++cc == p.C) {
            cc = 0;
            ++q;
            if (++px == p.P) {
                px = 0;
                q += row_skip;
                if (++py == p.P) { py = 0; q += frame_skip; }
            }
        }
        return v;
    }
};

__device__ __forceinline__ void patch_norm(const PatchP& p, int b, int l, float& mean, float& inv) {
    mean = 0.0f;
    inv = 1.0f;
    if (!p.norm_pix) return;
    double s = 0.0, q = 0.0;
    PatchWalk pw(p, b, l);
    for (int f = 0; f < p.PD; ++f) {
        const float v = pw.next(p);
        s += v;
        q += (double)v * v;
    }
    const double m = s / p.PD;
    const double var = (q - p.PD * m * m) / (p.PD - 1);   // unbiased, torch.var default (prithvi.py:343)
    mean = (float)m;
    inv = (float)(1.0 / sqrt(var + 1.0e-6));
}

// One workgroup = 256 consecutive columns (patches) of one sample x one (tt, py) row of the patches: lanes along the column
// read PRED / write DPRED rows coalesced ([PD][LP], row f = ((tt*P + py)*P + px)*C + c) and each lane reads its P-pixel patch
// row of every channel as float4s.  (One thread per whole patch - the first version - left 12.5 k threads walking 1,536
// strided elements each: 511 / 769 us for 150 MB.)
template <bool BWD, bool VEC>
__global__ void __launch_bounds__(256) mae_loss_rows_kernel(const PatchP p) {
    const int b = blockIdx.z, chunk = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    double num = 0.0, den = 0.0;
    if (col < p.LP) {
        const int l = col - p.L_OFF;
        const bool valid = l >= 0 && l < p.L;
        const float mk = valid ? p.mask[(int64_t)b * p.L + l] : 0.0f;
        const int tt = chunk / p.P, py = chunk - tt * p.P;
        const int64_t row0 = (int64_t)chunk * p.P * p.C;
        const float* pr = p.pred + ((int64_t)b * p.PD + row0) * p.LP + col;
        float* dst = BWD ? p.dpred + ((int64_t)b * p.PD + row0) * p.LP + col : nullptr;
        if (mk == 0.0f) {
            if (BWD)
                for (int r = 0; r < p.P * p.C; ++r) dst[(int64_t)r * p.LP] = 0.0f;
        } else {
            float mean, inv;
            patch_norm(p, b, l, mean, inv);
            const int w = l % p.gw, h = (l / p.gw) % p.gh, t = l / (p.gw * p.gh);
            const float* q = p.x + ((((int64_t)b * p.C) * p.T + (int64_t)t * p.TUB + tt) * p.H + (int64_t)h * p.P + py) * p.W + (int64_t)w * p.P;
            const int64_t cstride = (int64_t)p.T * p.H * p.W;
            float k = 0.0f;
            if (BWD) k = 2.0f * mk * (p.gout ? p.gout[0] : 1.0f) / ((float)p.PD * (float)p.acc[1]);
            float s = 0.0f;
            for (int c = 0; c < p.C; ++c) {
                const float* qc = q + (int64_t)c * cstride;
                if (VEC) {
                    for (int px = 0; px < p.P; px += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(qc + px);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int64_t r = ((int64_t)(px + j) * p.C + c) * p.LP;
                            const float d = pr[r] - (vv[j] - mean) * inv;
                            if (BWD) dst[r] = k * d; else s = fmaf(d, d, s);
                        }
                    }
                } else {
                    for (int px = 0; px < p.P; ++px) {
                        const int64_t r = ((int64_t)px * p.C + c) * p.LP;
                        const float d = pr[r] - (qc[px] - mean) * inv;
                        if (BWD) dst[r] = k * d; else s = fmaf(d, d, s);
                    }
                }
            }
            num = (double)(s / p.PD) * mk;
            if (chunk == 0) den = mk;
        }
    }
    if (!BWD) {
        // one atomic pair per workgroup (same-address f64 atomics execute one after the other at the memory side)
        __shared__ double red[8];
        num = wave_sum_d(num);
        den = wave_sum_d(den);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[wave] = num; red[4 + wave] = den; }
        __syncthreads();
        if (threadIdx.x == 0) {
            num = (red[0] + red[1]) + (red[2] + red[3]);
            den = (red[4] + red[5]) + (red[6] + red[7]);
            if (num != 0.0) atomic_add_d(p.acc, num);
            if (den != 0.0) atomic_add_d(p.acc + 1, den);
        }
    }
}

__global__ void mae_loss_finish_kernel(const PatchP p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p.loss[0] = (float)(p.acc[0] / p.acc[1]);   // 0/0 -> NaN at mask_ratio 0, as the reference
}

static bool mae_vec(const PatchP& p) {      // float4 patch rows: every row start is 16-byte aligned
    return p.P % 4 == 0 && p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
}

static int mae_loss_common(PatchP& p, const S2kOp& op) {
    if (int e = fill_patch(p, op.d)) return e;
    p.LP = op.d[7]; p.L_OFF = op.d[8]; p.norm_pix = op.d[9];
    if (p.LP < p.L + p.L_OFF || p.L_OFF < 0) { set_error("mae_loss: LP/L_OFF inconsistent with the patch grid"); return S2K_EINVAL; }
    if (p.B > 65535 || p.TUB * p.P > 65535) { set_error("mae_loss: batch / patch size beyond the launch grid"); return S2K_EINVAL; }
    return S2K_OK;
}

int launch_mae_loss_fwd(const S2kOp& op, const Ctx& c) {
    PatchP p{};
    if (int e = mae_loss_common(p, op)) return e;
    p.pred = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_FWD_T_PRED]);
    p.x = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_FWD_T_IMGS]);
    p.mask = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_FWD_T_MASK]);
    p.loss = ref_ptr<float>(c, op.t[S2K_MAE_LOSS_FWD_T_LOSS]);
    p.acc = ref_ptr<double>(c, op.t[S2K_MAE_LOSS_FWD_T_ACC]);
    CHECK_PTRS("mae_loss_fwd", p.pred, p.x, p.mask, p.loss, p.acc);
    if (!p.pred || !p.x || !p.mask || !p.loss || !p.acc) { set_error("mae_loss_fwd: missing tensor"); return S2K_EINVAL; }
    (void)hipMemsetAsync(p.acc, 0, 2 * sizeof(double), c.stream);
    const dim3 grid((unsigned)cdiv(p.LP, 256), (unsigned)(p.TUB * p.P), (unsigned)p.B);
    if (mae_vec(p)) hipLaunchKernelGGL((mae_loss_rows_kernel<false, true>), grid, dim3(256), 0, c.stream, p);
    else hipLaunchKernelGGL((mae_loss_rows_kernel<false, false>), grid, dim3(256), 0, c.stream, p);
    hipLaunchKernelGGL(mae_loss_finish_kernel, dim3(1), dim3(64), 0, c.stream, p);
    return S2K_OK;
}

int launch_mae_loss_bwd(const S2kOp& op, const Ctx& c) {
    PatchP p{};
    if (int e = mae_loss_common(p, op)) return e;
    p.pred = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_BWD_T_PRED]);
    p.x = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_BWD_T_IMGS]);
    p.mask = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_BWD_T_MASK]);
    p.acc = ref_ptr<double>(c, op.t[S2K_MAE_LOSS_BWD_T_ACC]);
    p.gout = ref_ptr<const float>(c, op.t[S2K_MAE_LOSS_BWD_T_GOUT]);
    p.dpred = ref_ptr<float>(c, op.t[S2K_MAE_LOSS_BWD_T_DPRED]);
    CHECK_PTRS("mae_loss_bwd", p.pred, p.x, p.mask, p.acc, p.gout, p.dpred);
    if (!p.pred || !p.x || !p.mask || !p.acc || !p.dpred) { set_error("mae_loss_bwd: missing tensor"); return S2K_EINVAL; }
    const dim3 grid((unsigned)cdiv(p.LP, 256), (unsigned)(p.TUB * p.P), (unsigned)p.B);
    if (mae_vec(p)) hipLaunchKernelGGL((mae_loss_rows_kernel<true, true>), grid, dim3(256), 0, c.stream, p);
    else hipLaunchKernelGGL((mae_loss_rows_kernel<true, false>), grid, dim3(256), 0, c.stream, p);
    return S2K_OK;
}

// ---------------- [B][C][L] -> [B][LOUT][C] through a 64x64 LDS tile -------------------------------------------
__global__ void __launch_bounds__(NTHREADS) transpose_cl_kernel(const float* x, float* y, int B, int C, int L, int l_off, int Lout, int ys,
                                                                int y_off) {
    __shared__ float tile[64][65];
    const int ct = blockIdx.x, lt = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // column tile ct covers Y columns [64 ct, 64 ct + 64) = source channels shifted by y_off
    for (int r = wave; r < 64; r += 4) {   // r = channel inside the tile, lanes along tokens
        const int cc = ct * 64 + r - y_off, j = lt * 64 + lane;
        tile[r][lane] = (cc >= 0 && cc < C && j < Lout) ? x[((int64_t)b * C + cc) * L + j + l_off] : 0.0f;
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {   // r = token inside the tile, lanes along channels
        const int j = lt * 64 + r, col = ct * 64 + lane;
        if (j < Lout && col < ys) y[((int64_t)b * Lout + j) * ys + col] = tile[lane][r];
    }
}

int launch_transpose_cl(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_TRANSPOSE_CL_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_TRANSPOSE_CL_T_Y]);
    CHECK_PTRS("transpose_cl", x, y);
    const int B = op.d[S2K_TRANSPOSE_CL_D_B], C = op.d[S2K_TRANSPOSE_CL_D_C], L = op.d[S2K_TRANSPOSE_CL_D_L];
    const int l_off = op.d[S2K_TRANSPOSE_CL_D_L_OFF], Lout = op.d[S2K_TRANSPOSE_CL_D_LOUT];
    int ys = op.d[S2K_TRANSPOSE_CL_D_YS], y_off = op.d[S2K_TRANSPOSE_CL_D_Y_OFF];
    if (ys <= 0) { ys = C; y_off = 0; }
    if (!x || !y || B <= 0 || C <= 0 || L <= 0 || l_off < 0 || Lout <= 0 || l_off + Lout > L || B > 65535 || y_off < 0 || y_off + C > ys) {
        set_error("transpose_cl: bad args"); return S2K_EINVAL;
    }
    hipLaunchKernelGGL(transpose_cl_kernel, dim3(cdiv(ys, 64), cdiv(Lout, 64), B), dim3(NTHREADS), 0, c.stream, x, y, B, C, L, l_off, Lout, ys, y_off);
    return S2K_OK;
}

// ---------------- decoder gather table from a caller-supplied ids_restore ------------------------------------------
__global__ void ids_to_dec_idx_kernel(const int64_t* ids, int* dec, int B, int L, int keep) {
    const int64_t n = (int64_t)B * (1 + L);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % (1 + L)), b = (int)(i / (1 + L));
        int v = 0;
        if (j > 0) {
            const int64_t r = ids[(int64_t)b * L + j - 1];
            v = (r >= 0 && r < keep) ? 1 + (int)r : -1;
        }
        dec[i] = v;
    }
}

int launch_ids_to_dec_idx(const S2kOp& op, const Ctx& c) {
    const int64_t* ids = ref_ptr<const int64_t>(c, op.t[S2K_IDS_TO_DEC_IDX_T_IDS]);
    int* dec = ref_ptr<int>(c, op.t[S2K_IDS_TO_DEC_IDX_T_DEC_IDX]);
    CHECK_PTRS("ids_to_dec_idx", ids, dec);
    const int B = op.d[S2K_IDS_TO_DEC_IDX_D_B], L = op.d[S2K_IDS_TO_DEC_IDX_D_L], keep = op.d[S2K_IDS_TO_DEC_IDX_D_KEEP];
    if (!ids || !dec || B <= 0 || L <= 0 || keep < 0) { set_error("ids_to_dec_idx: bad args"); return S2K_EINVAL; }
    const int64_t n = (int64_t)B * (1 + L);
    hipLaunchKernelGGL(ids_to_dec_idx_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, c.stream, ids, dec, B, L, keep);
    return S2K_OK;
}

}  // namespace s2k
