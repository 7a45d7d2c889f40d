// Multi-head softmax attention of the ViT blocks (timm `Attention` as used by prithvi.py:162-164,178-183:
// q,k,v = split(qkv(x)); softmax(q k^T * hd^-0.5) v) on the gfx950 f32 matrix cores, forward and backward.
//
// QKV is feature-major [B][3*H*hd][L]: one head's q / k / v are [hd][L] row-major tiles, which are exactly the
// LDS operand layouts of v_mfma_f32_32x32x2_f32 used here:
//   S[i][j]  = sum_d Q[d][i] K[d][j]      A(m=i,k=d) = Qs[d][i]  (lanes along i)   B(k=d,n=j) = Ks[d][j] (lanes along j)
//   O[d][i]  = sum_j V[d][j] P[i][j]      A(m=d,k=j) = Vs[d][j]  (row stride odd)  B(k=j,n=i) = Ps[i][j] (row stride odd)
// so nothing is transposed on the way in, and O / dQ / dK / dV come out feature-major with tokens on the lanes
// (128-B coalesced rows).  Rows of every LDS tile have an odd stride: a half-wave reading one column hits 32 banks.
// Softmax rows are reduced with wave shuffles; scores never leave LDS.
//
// forward:  one workgroup per (batch, head, 32-query tile).
// backward: one workgroup per (batch, head); it walks the query tiles, recomputes P, and keeps the dK / dV
//           accumulator tiles in registers (no atomics, deterministic).
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
    const float *qkv, *dout;
    float *o, *dqkv;
    int B, H, HD, L;
    int Lp, LS, hdp, hd2, ntq, mtiles;   // keys padded to 32, LDS row stride (odd), head dim padded to 32 / to even
    int red_ok;                          // backward: LDS has room for the dQ partial tiles
    float scale;
};

constexpr int QS = 33;   // row stride of the 32-query tiles

// acc += sum_k A(l31, k) * B(k, l31); element k of the operands at a[k * a_ks], b[k * b_ks]; K even
__device__ __forceinline__ void mfma_loop(f32x16& acc, const float* a, int a_ks, const float* b, int b_ks, int K, int lh) {
    for (int k2 = 0; k2 < K; k2 += 2) {
        const int k = k2 + lh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k * a_ks], b[k * b_ks], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
}

// rows [row0, row0 + nrows) of a [.][L] global matrix -> dst[r * stride + j], zero for r >= valid_rows or j >= L (j < Lp)
__device__ __forceinline__ void stage_rows(float* dst, int stride, const float* src, int L, int Lp, int nrows, int valid_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < nrows; r += 4)            // one row per wave and pass, lanes along the tokens: no index division
        for (int j = lane; j < Lp; j += 64) dst[r * stride + j] = (r < valid_rows && j < L) ? src[(int64_t)r * L + j] : 0.0f;
}

// 32 columns [i0, i0 + 32) of a [rows][L] matrix -> dst[r * QS + i]
__device__ __forceinline__ void stage_cols32(float* dst, const float* src, int L, int i0, int nrows, int valid_rows) {
    for (int e = threadIdx.x; e < nrows * 32; e += NTHREADS) {
        const int r = e >> 5, i = e & 31;
        dst[r * QS + i] = (r < valid_rows && i0 + i < L) ? src[(int64_t)r * L + i0 + i] : 0.0f;
    }
}

// scores of one query tile: Ss[i][j] = scale * <q_i, k_j>, -inf for j >= L; the waves split the 32-key column tiles
__device__ __forceinline__ void scores_tile(const AttnP& p, const float* Qs, const float* Ks, float* Ss, int wave, int l31, int lh) {
    const int ntiles = p.Lp >> 5;
    for (int nt = wave; nt < ntiles; nt += 4) {
        f32x16 acc;
        zero16(acc);
        mfma_loop(acc, Qs + l31, QS, Ks + 32 * nt + l31, p.LS, p.hd2, lh);
        const int j = 32 * nt + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
            Ss[i * p.LS + j] = j < p.L ? acc[r] * p.scale : -INFINITY;
        }
    }
}

// row softmax in place (wave w owns rows 8w .. 8w+7); padded key columns come out as exact zeros
__device__ __forceinline__ void softmax_rows(const AttnP& p, float* Ss, int wave, int lane) {
    for (int i = 8 * wave; i < 8 * wave + 8; ++i) {
        float* row = Ss + i * p.LS;
        float mx = -INFINITY;
        for (int j = lane; j < p.Lp; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.0f;
        for (int j = lane; j < p.Lp; j += 64) {
            const float e = j < p.L ? __expf(row[j] - mx) : 0.0f;   // v_exp_f32: ~1 ulp, far inside the 1e-3 bar
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < p.Lp; j += 64) row[j] *= inv;
    }
}

// ---------------- forward ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(NTHREADS) attn_fwd_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* KV = smem;                           // [hdp][LS]  K, then V
    float* Qs = KV + p.hdp * p.LS;              // [hdp][QS]
    float* Ss = Qs + p.hdp * QS;                // [32][LS]
    float* red = Ss + 32 * p.LS;                // [parts][hdp][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.x % p.ntq;
    const int h = (blockIdx.x / p.ntq) % p.H;
    const int b = blockIdx.x / (p.ntq * p.H);
    const int D = p.H * p.HD, i0 = qt * 32;
    const float* q = p.qkv + ((int64_t)b * 3 * D + h * p.HD) * p.L;
    const float* k = q + (int64_t)D * p.L;
    const float* v = k + (int64_t)D * p.L;

    stage_rows(KV, p.LS, k, p.L, p.Lp, p.hdp, p.HD);
    stage_cols32(Qs, q, p.L, i0, p.hdp, p.HD);
    __syncthreads();
    scores_tile(p, Qs, KV, Ss, wave, l31, lh);
    __syncthreads();
    stage_rows(KV, p.LS, v, p.L, p.Lp, p.hdp, p.HD);   // K is done: V takes its place while the softmax runs
    softmax_rows(p, Ss, wave, lane);
    __syncthreads();
    // O[d][i] = sum_j V[d][j] P[i][j]: (row tile mt, key range part) per wave, partial tiles summed through LDS
    const int parts = 4 / p.mtiles;
    const int mt = wave % p.mtiles, part = wave / p.mtiles;
    const int klen = p.Lp / parts;
    f32x16 acc;
    zero16(acc);
    mfma_loop(acc, KV + (32 * mt + l31) * p.LS + part * klen, 1, Ss + l31 * p.LS + part * klen, 1, klen, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        red[(part * p.hdp + d) * 32 + l31] = acc[r];
    }
    __syncthreads();
    float* o = p.o + ((int64_t)b * D + h * p.HD) * p.L;
    for (int e = threadIdx.x; e < p.hdp * 32; e += NTHREADS) {
        const int d = e >> 5, i = e & 31;
        if (d < p.HD && i0 + i < p.L) {
            float s = 0.0f;
            for (int pp = 0; pp < parts; ++pp) s += red[(pp * p.hdp + d) * 32 + i];
            o[(int64_t)d * p.L + i0 + i] = s;
        }
    }
}

// ---------------- backward -------------------------------------------------------------------------------
constexpr int MAX_TPW = 4;   // dK / dV accumulator tiles per wave

__global__ void __launch_bounds__(NTHREADS, 1) attn_bwd_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                           // [hdp][LS]
    float* Vs = Ks + p.hdp * p.LS;              // [hdp][LS]
    float* Qs = Vs + p.hdp * p.LS;              // [hdp][QS]
    float* dOs = Qs + p.hdp * QS;               // [hdp][QS]
    float* Ss = dOs + p.hdp * QS;               // [32][LS]   P, then dS
    float* dpart = Ss + 32 * p.LS;              // [4][32]    per-wave partial row sums of dP * P
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
    const int D = p.H * p.HD;
    const float* q = p.qkv + ((int64_t)b * 3 * D + h * p.HD) * p.L;
    const float* k = q + (int64_t)D * p.L;
    const float* v = k + (int64_t)D * p.L;
    const float* dO = p.dout + ((int64_t)b * D + h * p.HD) * p.L;
    float* dq = p.dqkv + ((int64_t)b * 3 * D + h * p.HD) * p.L;
    float* dk = dq + (int64_t)D * p.L;
    float* dv = dk + (int64_t)D * p.L;
    const int ntiles = p.Lp >> 5;
    const int ntot = p.mtiles * ntiles;         // dK / dV tiles (mt, nt), dealt round-robin to the waves

    f32x16 aK[MAX_TPW], aV[MAX_TPW];
#pragma unroll
    for (int t = 0; t < MAX_TPW; ++t) { zero16(aK[t]); zero16(aV[t]); }

    stage_rows(Ks, p.LS, k, p.L, p.Lp, p.hdp, p.HD);
    stage_rows(Vs, p.LS, v, p.L, p.Lp, p.hdp, p.HD);
    const int parts = 4 / p.mtiles;
    const int klen = p.Lp / parts;

    for (int qt = 0; qt < p.ntq; ++qt) {
        const int i0 = qt * 32;
        stage_cols32(Qs, q, p.L, i0, p.hdp, p.HD);
        stage_cols32(dOs, dO, p.L, i0, p.hdp, p.HD);
        __syncthreads();
        scores_tile(p, Qs, Ks, Ss, wave, l31, lh);
        __syncthreads();
        softmax_rows(p, Ss, wave, lane);
        __syncthreads();
        // dV[d][j] += sum_i dO[d][i] P[i][j]
#pragma unroll
        for (int t = 0; t < MAX_TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < ntot) {
                const int mt = tile % p.mtiles, nt = tile / p.mtiles;
                mfma_loop(aV[t], dOs + (32 * mt + l31) * QS, 1, Ss + 32 * nt + l31, p.LS, 32, lh);
            }
        }
        // dP[i][j] = sum_d dO[d][i] V[d][j] for this wave's key tiles; delta_i = sum_j dP * P
        f32x16 dP[2];
        float rowpart[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rowpart[r] = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            zero16(dP[u]);
            const int nt = wave + 4 * u;
            if (nt < ntiles) {
                mfma_loop(dP[u], dOs + l31, QS, Vs + 32 * nt + l31, p.LS, p.hd2, lh);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    rowpart[r] = fmaf(dP[u][r], Ss[i * p.LS + 32 * nt + l31], rowpart[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = half_sum_hi(rowpart[r]);
            if (l31 == 31) dpart[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = s;
        }
        __syncthreads();   // every wave is done reading P (dV above, delta here): Ss may be overwritten with dS
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nt = wave + 4 * u;
            if (nt < ntiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float delta = (dpart[i] + dpart[32 + i]) + (dpart[64 + i] + dpart[96 + i]);
                    float* s = Ss + i * p.LS + 32 * nt + l31;
                    *s = *s * (dP[u][r] - delta) * p.scale;
                }
            }
        }
        __syncthreads();
        // dK[d][j] += sum_i Q[d][i] dS[i][j]
#pragma unroll
        for (int t = 0; t < MAX_TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < ntot) {
                const int mt = tile % p.mtiles, nt = tile / p.mtiles;
                mfma_loop(aK[t], Qs + (32 * mt + l31) * QS, 1, Ss + 32 * nt + l31, p.LS, 32, lh);
            }
        }
        // dQ[d][i] = sum_j K[d][j] dS[i][j]: (row tile, key range) per wave; the key-range partials are combined in a
        // fixed order (part 0 stores, the others add after a barrier) so the result is reproducible
        {
            const int mt = wave % p.mtiles, part = wave / p.mtiles;
            f32x16 acc;
            zero16(acc);
            mfma_loop(acc, Ks + (32 * mt + l31) * p.LS + part * klen, 1, Ss + l31 * p.LS + part * klen, 1, klen, lh);
            if (p.red_ok) {   // LDS has room for the partial tiles: one barrier, fixed-order sum, coalesced store
                float* red = dpart + 128;                       // [parts][hdp][32]
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    red[(part * p.hdp + d) * 32 + l31] = acc[r];
                }
                __syncthreads();
                for (int e = threadIdx.x; e < p.hdp * 32; e += NTHREADS) {
                    const int d = e >> 5, i = e & 31;
                    if (d < p.HD && i0 + i < p.L) {
                        float sum = 0.0f;
                        for (int pp = 0; pp < parts; ++pp) sum += red[(pp * p.hdp + d) * 32 + i];
                        dq[(int64_t)d * p.L + i0 + i] = sum;
                    }
                }
                __syncthreads();
            } else
            for (int pp = 0; pp < parts; ++pp) {
                if (part == pp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (d < p.HD && i0 + l31 < p.L) {
                            float* dst = dq + (int64_t)d * p.L + i0 + l31;
                            if (pp == 0) *dst = acc[r];
                            else atomicAdd(dst, acc[r]);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MAX_TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile < ntot) {
            const int mt = tile % p.mtiles, nt = tile / p.mtiles;
            const int j = 32 * nt + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (d < p.HD && j < p.L) {
                    dk[(int64_t)d * p.L + j] = aK[t][r];
                    dv[(int64_t)d * p.L + j] = aV[t][r];
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
static int fill_attn(AttnP& p, const int32_t* d, float scale) {
    p.B = d[0]; p.H = d[1]; p.HD = d[2]; p.L = d[3];
    p.scale = scale;
    if (p.B <= 0 || p.H <= 0 || p.HD <= 0 || p.L <= 0 || p.HD > 64) { set_error("attention: unsupported dims (head dim <= 64)"); return S2K_EINVAL; }
    p.Lp = cdiv(p.L, 32) * 32;
    p.LS = p.Lp + 1;
    p.hdp = cdiv(p.HD, 32) * 32;
    p.hd2 = (p.HD + 1) & ~1;
    p.ntq = p.Lp / 32;
    p.mtiles = p.hdp / 32;
    return S2K_OK;
}

int launch_attn_fwd(const S2kOp& op, const Ctx& c) {
    AttnP p{};
    if (int e = fill_attn(p, op.d, op.f[S2K_ATTN_FWD_F_SCALE])) return e;
    p.qkv = ref_ptr<const float>(c, op.t[S2K_ATTN_FWD_T_QKV]);
    p.o = ref_ptr<float>(c, op.t[S2K_ATTN_FWD_T_O]);
    if (p.qkv == reinterpret_cast<const float*>(1) || p.o == reinterpret_cast<float*>(1)) { set_error("attn_fwd: null base"); return S2K_EFAULT; }
    if (!p.qkv || !p.o) { set_error("attn_fwd: missing tensor"); return S2K_EINVAL; }
    const size_t lds = ((size_t)p.hdp * p.LS + (size_t)p.hdp * QS + 32 * (size_t)p.LS + 128 * 32) * sizeof(float);
    if (lds > 160 * 1024) { set_error("attn_fwd: %d tokens x head dim %d needs %zu B of LDS (single-pass kernel, max 160 KB)", p.L, p.HD, lds); return S2K_EINVAL; }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const int64_t blocks = (int64_t)p.B * p.H * p.ntq;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)blocks), dim3(NTHREADS), lds, c.stream, p);
    return S2K_OK;
}

int launch_attn_bwd(const S2kOp& op, const Ctx& c) {
    AttnP p{};
    if (int e = fill_attn(p, op.d, op.f[S2K_ATTN_BWD_F_SCALE])) return e;
    p.qkv = ref_ptr<const float>(c, op.t[S2K_ATTN_BWD_T_QKV]);
    p.dout = ref_ptr<const float>(c, op.t[S2K_ATTN_BWD_T_DO]);
    p.dqkv = ref_ptr<float>(c, op.t[S2K_ATTN_BWD_T_DQKV]);
    if (p.qkv == reinterpret_cast<const float*>(1) || p.dout == reinterpret_cast<const float*>(1) || p.dqkv == reinterpret_cast<float*>(1)) {
        set_error("attn_bwd: null base"); return S2K_EFAULT;
    }
    if (!p.qkv || !p.dout || !p.dqkv) { set_error("attn_bwd: missing tensor"); return S2K_EINVAL; }
    if (p.mtiles * (p.Lp / 32) > 4 * MAX_TPW || p.Lp / 32 > 8) {
        set_error("attn_bwd: %d tokens x head dim %d exceeds the register-resident dK/dV tiles", p.L, p.HD); return S2K_EINVAL;
    }
    size_t lds = (2 * (size_t)p.hdp * p.LS + 2 * (size_t)p.hdp * QS + 32 * (size_t)p.LS + 128) * sizeof(float);
    if (lds > 160 * 1024) { set_error("attn_bwd: %d tokens x head dim %d needs %zu B of LDS (max 160 KB)", p.L, p.HD, lds); return S2K_EINVAL; }
    p.red_ok = lds + 128 * 32 * sizeof(float) <= 160 * 1024;
    if (p.red_ok) lds += 128 * 32 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(p.B * p.H)), dim3(NTHREADS), lds, c.stream, p);
    return S2K_OK;
}

}  // namespace s2k
