// Multi-head softmax attention of the ViT blocks (timm `Attention` as used by prithvi.py:162-164,178-183:
// q,k,v = split(qkv(x)); softmax(q k^T * hd^-0.5) v) on the gfx950 f32 matrix cores, forward and backward.
//
// QKV is feature-major [B][3*H*hd][LS] (LS = row stride >= L tokens): one head's q / k / v are [hd][LS] row-major
// tiles, and v_mfma_f32_32x32x2_f32 takes them straight from global memory, one element per lane:
//   streamed A(m = token, k = d) and held B(k = d, n = token) are 128-B coalesced row segments (lanes along tokens).
// Every wave is an independent unit - no LDS, no barriers, no atomics - so occupancy hides the load latency:
//   forward   unit = (batch, head, 32-query tile): S^T[j][i] = sum_d K[d][j] Q[d][i] per 32-key tile, with the QUERY on
//             the lanes, so the softmax row statistics are per-lane scalars (online softmax: running max / sum, the
//             output tile rescaled in place; one cross-half exchange per tile).  The probabilities never leave their
//             accumulator registers: P^T[j][i] is already the B operand (k = j, n = i) of O[d][i] += V[d][j] P^T[j][i]
//             if step s of the contraction takes key (s&3) + 8(s>>2) + 4*half - the accumulator's own row order - and
//             in that order the A operand V[d][j..j+3] is a 16-byte load per lane (lanes along d).
//             The per-query log-sum-exp is kept for the backward.
//   backward  two kernels of the same shape.  dQ: unit = (b, h, query tile) recomputes S^T and dP^T per key tile,
//             dS^T = P^T (dP^T - delta_i) with delta_i = sum_d dO[d][i] O[d][i] a per-lane scalar, dQ[d][i] += K[d][j] dS^T[j][i].
//             dK/dV: unit = (b, h, key tile) recomputes S and dP with the KEY on the lanes and walks the query tiles:
//             dV[d][j] += dO[d][i] P[i][j], dK[d][j] += Q[d][i] dS[i][j].  Recomputing the scores in both
//             orientations costs 7 tile products per (query tile, key tile) pair instead of 5, and removes every
//             cross-wave reduction (the results are deterministic).
// Tokens L..LS-1 (row padding) are written as zeros in O, dQKV, LSE.
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

constexpr int AW = 4;   // independent waves per workgroup

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

// acc[m = streamed token][n = held token] = sum_d T[d][tok] * held[d][n]   (rows d >= HD meet held zeros)
template <int MT>
__device__ __forceinline__ f32x16 tile_dd(rsrc_t rs, uint32_t row0, int tok, int LS, int lh, const float (&held)[16 * MT]) {
    float a[16 * MT];
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) a[s] = bload(rs, (row0 + (uint32_t)(2 * s + lh) * LS + tok) * 4u);
    f32x16 acc;
    zero16(acc);
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], held[s], acc, 0, 0, 0);
    return acc;
}

template <bool VEC>
__device__ __forceinline__ f32x4 load_t4(rsrc_t rs, uint32_t elem) {
    if (VEC) return bload4(rs, elem * 4u);
    f32x4 v;
    v[0] = bload(rs, elem * 4u);
    v[1] = bload(rs, elem * 4u + 4u);
    v[2] = bload(rs, elem * 4u + 8u);
    v[3] = bload(rs, elem * 4u + 12u);
    return v;
}

// acc[mt][m = d][n] += sum_t T[d][tok0 + t] * b[t][n], b = an accumulator tile (rows t in register order)
template <int MT, bool VEC>
__device__ __forceinline__ void tile_acc(f32x16 (&acc)[MT], rsrc_t rs, uint32_t row0, int tok0, int LS, int l31, int lh, const f32x16& b) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const uint32_t base = row0 + (uint32_t)(32 * mt + l31) * LS + tok0 + 4 * lh;
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = load_t4<VEC>(rs, base + 8 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g][c], b[4 * g + c], acc[mt], 0, 0, 0);
    }
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

struct Unit { int b, h, tile; int64_t bh; bool ok; };
__device__ __forceinline__ Unit unit_of(const AttnP& p) {
    Unit u;
    const int64_t id = (int64_t)blockIdx.x * AW + (threadIdx.x >> 6);
    u.tile = (int)(id % p.nt);
    u.bh = id / p.nt;
    u.ok = u.bh < (int64_t)p.B * p.H;
    u.h = (int)(u.bh % p.H);
    u.b = (int)(u.bh / p.H);
    return u;
}

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

// ---------------- forward ---------------------------------------------------------------------------------
template <int MT, bool VEC>
__global__ void __launch_bounds__(64 * AW, 2) attn_fwd_kernel(const AttnP p) {
    const Unit u = unit_of(p);
    if (!u.ok) return;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int i = 32 * u.tile + l31;
    float qreg[16 * MT];
    load_held<MT>(qreg, rs, qo, min(i, L - 1), p.HD, LS, lh, p.scale);
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(acc[mt]);
    float m = -INFINITY, l = 0.0f;
    for (int kt = 0; kt < p.nt; ++kt) {
        f32x16 s = tile_dd<MT>(rs, ko, min(32 * kt + l31, L - 1), LS, lh, qreg);   // rows: key, columns: query
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = 32 * kt + mrow(r, lh) < L ? s[r] : -INFINITY;
            tm = fmaxf(tm, s[r]);
        }
        tm = fmaxf(tm, xhalf(tm));
        const float mn = fmaxf(m, tm);            // finite: every tile holds at least one real key
        const float corr = __expf(m - mn);
        float ts = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - mn);             // v_exp_f32: ~1 ulp, far inside the 1e-3 bar
            ts += s[r];
        }
        ts += xhalf(ts);
        l = l * corr + ts;
        m = mn;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] *= corr;
        tile_acc<MT, VEC>(acc, rs, vo, 32 * kt, LS, l31, lh, s);
    }
    store_rows<MT>(p.o + ((int64_t)u.b * D + u.h * p.HD) * LS, acc, i, lh, p, 1.0f / l);
    if (lh == 0 && i < LS) p.lse[u.bh * LS + i] = i < L ? m + __logf(l) : 0.0f;
}

// ---------------- backward -------------------------------------------------------------------------------
template <int MT, bool VEC>
__global__ void __launch_bounds__(64 * AW, 2) attn_bwd_dq_kernel(const AttnP p) {
    const Unit u = unit_of(p);
    if (!u.ok) return;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const rsrc_t rdo = make_rsrc(p.dout + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const rsrc_t ro = make_rsrc(p.o_in + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int i = 32 * u.tile + l31, ic = min(i, L - 1);
    float qreg[16 * MT], doreg[16 * MT];
    load_held<MT>(qreg, rs, qo, ic, p.HD, LS, lh, p.scale);
    load_held<MT>(doreg, rdo, qo, ic, p.HD, LS, lh, 1.0f);
    float dl = 0.0f;                              // delta_i = sum_d dO[d][i] O[d][i] = sum_j P[i][j] dP[i][j]
#pragma unroll
    for (int s = 0; s < 16 * MT; ++s) dl = fmaf(doreg[s], bload(ro, (qo + (uint32_t)min(2 * s + lh, p.HD - 1) * LS + ic) * 4u), dl);
    dl += xhalf(dl);
    const float lse = p.lse[u.bh * LS + ic];
    if (lh == 0 && i < LS) p.delta[u.bh * LS + i] = i < L ? dl : 0.0f;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(acc[mt]);
    for (int kt = 0; kt < p.nt; ++kt) {
        const int jc = min(32 * kt + l31, L - 1);
        f32x16 s = tile_dd<MT>(rs, ko, jc, LS, lh, qreg);            // S^T[j][i] (scaled)
        const f32x16 dp = tile_dd<MT>(rs, vo, jc, LS, lh, doreg);    // dP^T[j][i] = sum_d V[d][j] dO[d][i]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = 32 * kt + mrow(r, lh) < L ? __expf(s[r] - lse) : 0.0f;
            s[r] = pr * (dp[r] - dl);
        }
        tile_acc<MT, VEC>(acc, rs, ko, 32 * kt, LS, l31, lh, s);
    }
    store_rows<MT>(p.dqkv + ((int64_t)u.b * 3 * D + u.h * p.HD) * LS, acc, i, lh, p, p.scale);
}

template <int MT, bool VEC>
__global__ void __launch_bounds__(64 * AW, 2) attn_bwd_dkv_kernel(const AttnP p) {
    const Unit u = unit_of(p);
    if (!u.ok) return;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int D = p.H * p.HD, LS = p.LS, L = p.L;
    const rsrc_t rs = make_rsrc(p.qkv + (int64_t)u.b * 3 * D * LS, (int64_t)3 * D * LS * 4);
    const rsrc_t rdo = make_rsrc(p.dout + (int64_t)u.b * D * LS, (int64_t)D * LS * 4);
    const uint32_t qo = (uint32_t)u.h * p.HD * LS, ko = qo + (uint32_t)D * LS, vo = ko + (uint32_t)D * LS;
    const int j = 32 * u.tile + l31, jc = min(j, L - 1);
    float kreg[16 * MT], vreg[16 * MT];
    load_held<MT>(kreg, rs, ko, jc, p.HD, LS, lh, p.scale);
    load_held<MT>(vreg, rs, vo, jc, p.HD, LS, lh, 1.0f);
    const float* lse = p.lse + u.bh * LS;
    const float* delta = p.delta + u.bh * LS;
    f32x16 aK[MT], aV[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { zero16(aK[mt]); zero16(aV[mt]); }
    for (int qt = 0; qt < p.nt; ++qt) {
        const int ic = min(32 * qt + l31, L - 1);
        f32x16 s = tile_dd<MT>(rs, qo, ic, LS, lh, kreg);             // S[i][j] (scaled): rows query, columns key
        f32x16 dp = tile_dd<MT>(rdo, qo, ic, LS, lh, vreg);           // dP[i][j] = sum_d dO[d][i] V[d][j]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = 32 * qt + mrow(r, lh), iic = min(ii, L - 1);
            const float pr = ii < L ? __expf(s[r] - lse[iic]) : 0.0f;
            s[r] = pr;
            dp[r] = pr * (dp[r] - delta[iic]);
        }
        tile_acc<MT, VEC>(aV, rdo, qo, 32 * qt, LS, l31, lh, s);
        tile_acc<MT, VEC>(aK, rs, qo, 32 * qt, LS, l31, lh, dp);
    }
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
static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

#define ATTN_LAUNCH(kernel, p, vec, stream)                                                                          \
    do {                                                                                                             \
        const int64_t units = (int64_t)(p).B * (p).H * (p).nt;                                                       \
        const dim3 grid((unsigned)cdiv64(units, AW)), block(64 * AW);                                                \
        if ((p).HD <= 32) {                                                                                          \
            if (vec) hipLaunchKernelGGL((kernel<1, true>), grid, block, 0, stream, p);                               \
            else hipLaunchKernelGGL((kernel<1, false>), grid, block, 0, stream, p);                                  \
        } else {                                                                                                     \
            if (vec) hipLaunchKernelGGL((kernel<2, true>), grid, block, 0, stream, p);                               \
            else hipLaunchKernelGGL((kernel<2, false>), grid, block, 0, stream, p);                                  \
        }                                                                                                            \
    } while (0)

int launch_attn_fwd(const S2kOp& op, const Ctx& c) {
    AttnP p{};
    if (int e = fill_attn(p, op.d, op.f[S2K_ATTN_FWD_F_SCALE])) return e;
    p.qkv = ref_ptr<const float>(c, op.t[S2K_ATTN_FWD_T_QKV]);
    p.o = ref_ptr<float>(c, op.t[S2K_ATTN_FWD_T_O]);
    p.lse = ref_ptr<float>(c, op.t[S2K_ATTN_FWD_T_LSE]);
    if (bad(p.qkv) || bad(p.o) || bad(p.lse)) { set_error("attn_fwd: missing tensor or null base"); return S2K_EINVAL; }
    const bool vec = p.LS % 4 == 0 && aligned16(p.qkv);
    ATTN_LAUNCH(attn_fwd_kernel, p, vec, c.stream);
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
    const bool vec = p.LS % 4 == 0 && aligned16(p.qkv) && aligned16(p.dout);
    ATTN_LAUNCH(attn_bwd_dq_kernel, p, vec, c.stream);      // also writes delta, which the dK/dV kernel reads
    ATTN_LAUNCH(attn_bwd_dkv_kernel, p, vec, c.stream);
    return S2K_OK;
}

}  // namespace s2k
