// Multi-head softmax attention of the ViT blocks (timm `Attention` as used by prithvi.py:162-164,178-183:
// q,k,v = split(qkv(x)); softmax(q k^T * hd^-0.5) v) on the gfx950 f32 matrix cores, forward and backward.
//
// QKV is feature-major [B][3*H*hd][LS] (LS = row stride >= L tokens): one head's q / k / v are [hd][LS] row-major tiles, i.e.
// already the operand layouts of v_mfma_f32_32x32x2_f32 (lanes along tokens).  A workgroup = 4 waves = 4 consecutive
// 32-token tiles of one (batch, head); each wave keeps ITS tile's operand in registers (one element per lane, coalesced
// loads) and all four stream the other side's tiles, which are fetched once per workgroup into a double-buffered LDS tile
// (registers first, so the loads fly during the MFMAs; one barrier per tile).
//   forward   wave = 32-query tile: S^T[j][i] = sum_d K[d][j] Q[d][i] per 32-key tile with the QUERY on the lanes, so the
//             softmax row statistics are per-lane scalars (online softmax: running max / sum, the output tile rescaled in
//             place; one cross-half exchange per tile).  The probabilities never leave their accumulator registers:
//             P^T[j][i] is already the B operand (k = j, n = i) of O[d][i] += V[d][j] P^T[j][i] if step s of the contraction
//             takes key (s&3) + 8(s>>2) + 4*half - the accumulator's own row order.  The per-query log-sum-exp is kept
//             for the backward.
//   backward  two kernels of the same shape.  dQ: wave = query tile, recomputes S^T and dP^T per key tile,
//             dS^T = P^T (dP^T - delta_i) with delta_i = sum_d dO[d][i] O[d][i] a per-lane scalar, dQ[d][i] += K[d][j] dS^T[j][i].
//             dK/dV: wave = key tile, recomputes S and dP with the KEY on the lanes and walks the query tiles:
//             dV[d][j] += dO[d][i] P[i][j], dK[d][j] += Q[d][i] dS[i][j].  Recomputing the scores in both orientations
//             costs 7 tile products per (query tile, key tile) pair instead of 5 and removes every cross-wave reduction
//             (no atomics: the results are deterministic).
// Tokens L..LS-1 (row padding) are written as zeros in O, dQKV, LSE.  Any L; head dim <= 64.
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

struct AttnP {
    const float *qkv, *dout, *o_in;
    float *o, *dqkv, *lse, *delta;
    int B, H, HD, L, LS;
    int nt;       // 32-token tiles
    float scale;
};

constexpr int AW = 4;   // waves (= 32-token tiles of one (batch, head)) per workgroup

__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
}
// row (M index) of accumulator register r in lane half lh of a 32x32 tile; the column is lane & 31
__device__ __forceinline__ int mrow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// held B operand (k = d = 2s + lh, n = token): reg[s] = mul * T[d][tok], zero for d >= HD
template <int MT>
__device__ __forceinline__ void load_held(float (&reg)[16 * MT], rsrc_t rs, uint32_t row0, int tok, int HD, int LS, int lh, float mul) {
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) {
        const int d = 2 * s + lh;
        const float x = bload(rs, (row0 + (uint32_t)min(d, HD - 1) * LS + tok) * 4u);
        reg[s] = x * (d < HD ? mul : 0.0f);
    }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// store acc rows d < HD of lane column tok: value * mul for tok < L, zero for the row padding
template <int MT>
__device__ __forceinline__ void store_rows(float* dst, const f32x16 (&acc)[MT], int tok, int lh, const AttnP& p, float mul) {
    if (tok >= p.LS) return;
    const float keep = tok < p.L ? mul : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * mt + mrow(r, lh);
            if (d < p.HD) dst[(int64_t)d * p.LS + tok] = acc[mt][r] * keep;
        }
}

// ---------------- streamed tiles shared by the workgroup through LDS ---------------------------------------------
// The four waves of a workgroup own four consecutive tiles of the SAME (batch, head) and stream the same K / V (forward, dQ)
// or Q / dO (dK/dV) tiles: each [hd][32-token] tile is fetched ONCE per workgroup (coalesced rows, registers first so the
// loads fly while the waves compute, then LDS, double-buffered: one barrier per tile) instead of once per wave from L2.
constexpr int TS = 33;   // LDS row stride of a staged [d][32 tokens] tile (odd: a half-wave reading a column hits 32 banks)

template <int MT>
struct Stage {
    float r[4 * MT];
    __device__ __forceinline__ void load(rsrc_t rs, uint32_t row0, int tok0, int HD, int LS, int L) {
        const int j = threadIdx.x & 31, d0 = threadIdx.x >> 5;       // 8 rows per pass, lanes along the tokens
        const uint32_t tok = (uint32_t)min(tok0 + j, L - 1);
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) r[i] = bload(rs, (row0 + (uint32_t)min(d0 + 8 * i, HD - 1) * LS + tok) * 4u);
    }
    __device__ __forceinline__ void store(float* T) const {
        const int j = threadIdx.x & 31, d0 = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) T[(d0 + 8 * i) * TS + j] = r[i];
    }
};

template <int MT>
__device__ __forceinline__ f32x16 tile_dd_lds(const float* T, int l31, int lh, const float (&held)[16 * MT]) {
    float a[16 * MT];
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) a[s] = T[(2 * s + lh) * TS + l31];
    f32x16 acc;
    zero16(acc);
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], held[s], acc, 0, 0, 0);
    return acc;
}

template <int MT>
__device__ __forceinline__ void tile_acc_lds(f32x16 (&acc)[MT], const float* T, int l31, int lh, const f32x16& b) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float* row = T + (32 * mt + l31) * TS + 4 * lh;
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * g + c] = row[8 * g + c];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[k], b[k], acc[mt], 0, 0, 0);
    }
}

struct Group { int b, h, own; int64_t bh; bool active; };
__device__ __forceinline__ Group group_of(const AttnP& p) {
    Group u;
    const int G = (p.nt + AW - 1) / AW;
    u.bh = blockIdx.x / G;
    u.own = (int)(blockIdx.x % G) * AW + (threadIdx.x >> 6);
    u.active = u.own < p.nt;
    u.h = (int)(u.bh % p.H);
    u.b = (int)(u.bh / p.H);
    return u;
}

template <int MT>
__global__ void __launch_bounds__(64 * AW, 2) attn_fwd_lds_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32 * MT * TS;
    const Group u = group_of(p);
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int i = 32 * u.own + l31;
    float qreg[16 * MT];
    load_held<MT>(qreg, rs, qo, min(i, L - 1), p.HD, LS, lh, p.scale);
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(acc[mt]);
    float m = -INFINITY, l = 0.0f;
    Stage<MT> sk, sv;
    sk.load(rs, ko, 0, p.HD, LS, L);
    sv.load(rs, vo, 0, p.HD, LS, L);
    sk.store(smem);
    sv.store(smem + TILE);
    __syncthreads();
    for (int kt = 0; kt < p.nt; ++kt) {
        const float* K = smem + (kt & 1) * 2 * TILE;
        const float* V = K + TILE;
        const bool more = kt + 1 < p.nt;
        if (more) {
            sk.load(rs, ko, 32 * (kt + 1), p.HD, LS, L);
            sv.load(rs, vo, 32 * (kt + 1), p.HD, LS, L);
        }
        if (u.active) {
            f32x16 s = tile_dd_lds<MT>(K, l31, lh, qreg);
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = 32 * kt + mrow(r, lh) < L ? s[r] : -INFINITY;
                tm = fmaxf(tm, s[r]);
            }
            tm = fmaxf(tm, xhalf(tm));
            const float mn = fmaxf(m, tm);
            const float corr = __expf(m - mn);
            float ts = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __expf(s[r] - mn);
                ts += s[r];
            }
            ts += xhalf(ts);
            l = l * corr + ts;
            m = mn;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] *= corr;
            tile_acc_lds<MT>(acc, V, l31, lh, s);
        }
        if (more) {
            float* nK = smem + ((kt + 1) & 1) * 2 * TILE;
            sk.store(nK);
            sv.store(nK + TILE);
        }
        __syncthreads();
    }
    if (!u.active) return;
    store_rows<MT>(p.o + ((int64_t)u.b * D + u.h * p.HD) * LS, acc, i, lh, p, 1.0f / l);
    if (lh == 0 && i < LS) p.lse[u.bh * LS + i] = i < L ? m + __logf(l) : 0.0f;
}

template <int MT>
__global__ void __launch_bounds__(64 * AW, 2) attn_bwd_dq_lds_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32 * MT * TS;
    const Group u = group_of(p);
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const rsrc_t rdo = make_rsrc(p.dout + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const rsrc_t ro = make_rsrc(p.o_in + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int i = 32 * u.own + l31, ic = min(i, L - 1);
    float qreg[16 * MT], doreg[16 * MT];
    load_held<MT>(qreg, rs, qo, ic, p.HD, LS, lh, p.scale);
    load_held<MT>(doreg, rdo, qo, ic, p.HD, LS, lh, 1.0f);
    float dl = 0.0f;
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) dl = fmaf(doreg[s], bload(ro, (qo + (uint32_t)min(2 * s + lh, p.HD - 1) * LS + ic) * 4u), dl);
    dl += xhalf(dl);
    const float lse = p.lse[u.bh * LS + ic];
    if (u.active && lh == 0 && i < LS) p.delta[u.bh * LS + i] = i < L ? dl : 0.0f;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(acc[mt]);
    Stage<MT> sk, sv;
    sk.load(rs, ko, 0, p.HD, LS, L);
    sv.load(rs, vo, 0, p.HD, LS, L);
    sk.store(smem);
    sv.store(smem + TILE);
    __syncthreads();
    for (int kt = 0; kt < p.nt; ++kt) {
        const float* K = smem + (kt & 1) * 2 * TILE;
        const float* V = K + TILE;
        const bool more = kt + 1 < p.nt;
        if (more) {
            sk.load(rs, ko, 32 * (kt + 1), p.HD, LS, L);
            sv.load(rs, vo, 32 * (kt + 1), p.HD, LS, L);
        }
        if (u.active) {
            f32x16 s = tile_dd_lds<MT>(K, l31, lh, qreg);
            const f32x16 dp = tile_dd_lds<MT>(V, l31, lh, doreg);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = 32 * kt + mrow(r, lh) < L ? __expf(s[r] - lse) : 0.0f;
                s[r] = pr * (dp[r] - dl);
            }
            tile_acc_lds<MT>(acc, K, l31, lh, s);
        }
        if (more) {
            float* nK = smem + ((kt + 1) & 1) * 2 * TILE;
            sk.store(nK);
            sv.store(nK + TILE);
        }
        __syncthreads();
    }
    if (!u.active) return;
    store_rows<MT>(p.dqkv + ((int64_t)u.b * 3 * D + u.h * p.HD) * LS, acc, i, lh, p, p.scale);
}

template <int MT>
__global__ void __launch_bounds__(64 * AW, 2) attn_bwd_dkv_lds_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32 * MT * TS;
    float* aux = smem + 4 * TILE;                      // [2 buffers][64]: lse | delta of the streamed query tile
    const Group u = group_of(p);
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const rsrc_t rdo = make_rsrc(p.dout + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int j = 32 * u.own + l31, jc = min(j, L - 1);
    float kreg[16 * MT], vreg[16 * MT];
    load_held<MT>(kreg, rs, ko, jc, p.HD, LS, lh, p.scale);
    load_held<MT>(vreg, rs, vo, jc, p.HD, LS, lh, 1.0f);
    const float* lse = p.lse + u.bh * LS;
    const float* delta = p.delta + u.bh * LS;
    f32x16 aK[MT], aV[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { zero16(aK[mt]); zero16(aV[mt]); }
    Stage<MT> sq, sd;
    float ax = 0.0f;
    auto load_aux = [&](int qt) {
        if (threadIdx.x < 64) {
            const int t = threadIdx.x & 31, tok = min(32 * qt + t, L - 1);
            ax = threadIdx.x < 32 ? lse[tok] : delta[tok];
        }
    };
    sq.load(rs, qo, 0, p.HD, LS, L);
    sd.load(rdo, qo, 0, p.HD, LS, L);
    load_aux(0);
    sq.store(smem);
    sd.store(smem + TILE);
    if (threadIdx.x < 64) aux[threadIdx.x] = ax;
    __syncthreads();
    for (int qt = 0; qt < p.nt; ++qt) {
        const float* Q = smem + (qt & 1) * 2 * TILE;
        const float* dO = Q + TILE;
        const float* A = aux + (qt & 1) * 64;
        const bool more = qt + 1 < p.nt;
        if (more) {
            sq.load(rs, qo, 32 * (qt + 1), p.HD, LS, L);
            sd.load(rdo, qo, 32 * (qt + 1), p.HD, LS, L);
            load_aux(qt + 1);
        }
        if (u.active) {
            f32x16 s = tile_dd_lds<MT>(Q, l31, lh, kreg);       // S[i][j] (scaled): rows query, columns key
            f32x16 dp = tile_dd_lds<MT>(dO, l31, lh, vreg);     // dP[i][j]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = mrow(r, lh);
                const float pr = 32 * qt + il < L ? __expf(s[r] - A[il]) : 0.0f;
                s[r] = pr;
                dp[r] = pr * (dp[r] - A[32 + il]);
            }
            tile_acc_lds<MT>(aV, dO, l31, lh, s);
            tile_acc_lds<MT>(aK, Q, l31, lh, dp);
        }
        if (more) {
            float* nQ = smem + ((qt + 1) & 1) * 2 * TILE;
            sq.store(nQ);
            sd.store(nQ + TILE);
            if (threadIdx.x < 64) aux[((qt + 1) & 1) * 64 + threadIdx.x] = ax;
        }
        __syncthreads();
    }
    if (!u.active) return;
    float* dk = p.dqkv + ((int64_t)u.b * 3 * D + D + u.h * p.HD) * LS;
    store_rows<MT>(dk, aK, j, lh, p, p.scale);
    store_rows<MT>(dk + (int64_t)D * LS, aV, j, lh, p, 1.0f);
}

// -------------------------------------------------------------------------------------------------------------
static int fill_attn(AttnP& p, const int32_t* d, float scale) {
    p.B = d[0]; p.H = d[1]; p.HD = d[2]; p.L = d[3]; p.LS = d[4] > 0 ? d[4] : d[3];
    p.scale = scale;
    if (p.B <= 0 || p.H <= 0 || p.HD <= 0 || p.L <= 0 || p.HD > 64 || p.LS < p.L) {
        set_error("attention: unsupported dims (head dim <= 64, row stride >= tokens)"); return S2K_EINVAL;
    }
    if ((int64_t)3 * p.H * p.HD * p.LS * 4 + 64ll * p.LS * 4 > 0x7ffffff0ll) { set_error("attention: one image's qkv exceeds 2 GiB"); return S2K_EINVAL; }
    p.nt = cdiv(p.L, 32);
    return S2K_OK;
}

static bool bad(const void* q) { return q == nullptr || q == reinterpret_cast<const void*>(1); }


// one workgroup per (batch, head, group of 4 tiles); LDS = ntiles_buf double-buffered [32*MT][TS] tiles + extra floats
#define ATTN_LAUNCH_LDS(kernel, p, ntiles_buf, extra, stream)                                                        \
    do {                                                                                                             \
        const int64_t groups = (int64_t)(p).B * (p).H * (((p).nt + AW - 1) / AW);                                    \
        const dim3 grid((unsigned)groups), block(64 * AW);                                                           \
        if ((p).HD <= 32) hipLaunchKernelGGL((kernel<1>), grid, block, ((ntiles_buf) * 32 * 1 * TS + (extra)) * sizeof(float), stream, p); \
        else hipLaunchKernelGGL((kernel<2>), grid, block, ((ntiles_buf) * 32 * 2 * TS + (extra)) * sizeof(float), stream, p); \
    } while (0)

int launch_attn_fwd(const S2kOp& op, const Ctx& c) {
    AttnP p{};
    if (int e = fill_attn(p, op.d, op.f[S2K_ATTN_FWD_F_SCALE])) return e;
    p.qkv = ref_ptr<const float>(c, op.t[S2K_ATTN_FWD_T_QKV]);
    p.o = ref_ptr<float>(c, op.t[S2K_ATTN_FWD_T_O]);
    p.lse = ref_ptr<float>(c, op.t[S2K_ATTN_FWD_T_LSE]);
    if (bad(p.qkv) || bad(p.o) || bad(p.lse)) { set_error("attn_fwd: missing tensor or null base"); return S2K_EINVAL; }
    ATTN_LAUNCH_LDS(attn_fwd_lds_kernel, p, 4, 0, c.stream);
    return S2K_OK;
}

int launch_attn_bwd(const S2kOp& op, const Ctx& c) {
    AttnP p{};
    if (int e = fill_attn(p, op.d, op.f[S2K_ATTN_BWD_F_SCALE])) return e;
    p.qkv = ref_ptr<const float>(c, op.t[S2K_ATTN_BWD_T_QKV]);
    p.dout = ref_ptr<const float>(c, op.t[S2K_ATTN_BWD_T_DO]);
    p.dqkv = ref_ptr<float>(c, op.t[S2K_ATTN_BWD_T_DQKV]);
    p.o_in = ref_ptr<const float>(c, op.t[S2K_ATTN_BWD_T_O]);
    p.lse = ref_ptr<float>(c, op.t[S2K_ATTN_BWD_T_LSE]);
    p.delta = ref_ptr<float>(c, op.t[S2K_ATTN_BWD_T_DELTA]);
    if (bad(p.qkv) || bad(p.dout) || bad(p.dqkv) || bad(p.o_in) || bad(p.lse) || bad(p.delta)) {
        set_error("attn_bwd: missing tensor or null base"); return S2K_EINVAL;
    }
    ATTN_LAUNCH_LDS(attn_bwd_dq_lds_kernel, p, 4, 0, c.stream);      // also writes delta, which the dK/dV kernel reads
    ATTN_LAUNCH_LDS(attn_bwd_dkv_lds_kernel, p, 4, 128, c.stream);
    return S2K_OK;
}

}  // namespace s2k
