// HBM-bound stages around the convolutions: BatchNorm finalize / backward (reduce, finalize, apply),
// squeeze-excitation (pool, FCs, backward), residual + drop-connect, channel sums, grad layout fold.
//
// Reference arithmetic being replaced: nn.BatchNorm2d train/eval (efficientnet_unet.py:170-175,
// 194,329,345,367), SE branch + x*sigmoid(SE(x)) (:346-381), _drop_connect + residual (:383-398).
// Work decomposition: one wave per (plane, chunk) task, float4 accesses when the plane size allows,
// wave shuffle reductions, one f64 atomic per task for per-channel sums.
#include "common.h"

namespace s2k {

constexpr int CHUNK = 4096;   // elements of one (b,c) plane handled by one wave

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

struct Task {
    int64_t plane;
    int start, count;
};

__device__ __forceinline__ bool get_task(int HW, int64_t nplanes, Task& t) {
    const int chunks = (HW + CHUNK - 1) / CHUNK;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= nplanes * chunks) return false;
    // (tasks fit 32 bits - task_blocks() checks - and a 64-bit division is ~150 instructions per wave)
    t.plane = (int64_t)((uint32_t)task / (uint32_t)chunks);
    const int ck = (int)(task - t.plane * chunks);
    t.start = ck * CHUNK;
    t.count = min(CHUNK, HW - t.start);
    return true;
}

static unsigned task_blocks(int HW, int64_t nplanes) {
    const int chunks = (HW + CHUNK - 1) / CHUNK;
    if (nplanes * chunks >= 0xffffffffll) return 0;          // (a launch of 0 blocks fails loudly in check_launch)
    return (unsigned)cdiv64(nplanes * chunks, 4);
}

// ---------------- AXPY ---------------------------------------------------------------------------
__global__ void axpy_kernel(const float* x, float* y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = reinterpret_cast<float4*>(y)[i];
        b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
        reinterpret_cast<float4*>(y)[i] = b;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += x[i];
}

int launch_axpy(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_AXPY_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_AXPY_T_Y]);
    CHECK_PTRS("axpy", x, y);
    const int64_t n = op.n[S2K_AXPY_N_COUNT];
    if (!x || !y || n <= 0) { set_error("axpy: bad args"); return S2K_EINVAL; }
    const int blocks = (int)std::min<int64_t>(cdiv64(n, 4 * 256), 2048);
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, c.stream, x, y, n);
    return S2K_OK;
}

// ---------------- WGRAD_FINALIZE -------------------------------------------------------------------
// table rows: {float offset, M, C, T, start}; thread e handles output element e of the concatenated
// conv-weight index space: GRADS[off + (m*C + c)*T + t] += WGS[off + (t*M + m)*C + c]
__global__ void wgrad_finalize_kernel(const int* table, int n_entries, const float* wgs, float* grads, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int lo = 0, hi = n_entries - 1;
        while (lo < hi) {  // last entry with start <= e
            const int mid = (lo + hi + 1) >> 1;
            if ((int64_t)table[mid * 5 + 4] <= e) lo = mid; else hi = mid - 1;
        }
        const int* r = table + lo * 5;
        const int64_t off = r[0];
        const int M = r[1], C = r[2], T = r[3];
        const int local = (int)(e - r[4]);
        const int t = local % T;
        const int mc = local / T;
        const int cc = mc % C, m = mc / C;
        grads[off + local] += wgs[off + ((int64_t)t * M + m) * C + cc];
    }
}

int launch_wgrad_finalize(const S2kOp& op, const Ctx& c) {
    const int* table = ref_ptr<const int>(c, op.t[S2K_WGRAD_FINALIZE_T_TABLE]);
    const float* wgs = ref_ptr<const float>(c, op.t[S2K_WGRAD_FINALIZE_T_WGS]);
    float* grads = ref_ptr<float>(c, op.t[S2K_WGRAD_FINALIZE_T_GRADS]);
    CHECK_PTRS("wgrad_finalize", table, wgs, grads);
    const int64_t total = op.n[S2K_WGRAD_FINALIZE_N_TOTAL];
    const int n = op.d[S2K_WGRAD_FINALIZE_D_N_ENTRIES];
    if (!table || !wgs || !grads || n <= 0 || total <= 0 || total > 0x7fffffff) { set_error("wgrad_finalize: bad args"); return S2K_EINVAL; }
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, c.stream, table, n, wgs, grads, total);
    return S2K_OK;
}

// ---------------- WEIGHT_PACK ----------------------------------------------------------------------
// table rows: {src_off, dst_off, M, K, T, s_m, s_k, s_t, flip, MP, KP, start}.  One workgroup = one 64 (k,tap) x 64 (m)
// tile of one entry (MP % 128 == 0, KP % 64 == 0, so tiles never straddle entries and `start` is a multiple of 4096).
// Forward packs read rows of [M][K*T] weights (contiguous along (k,tap)) and write [K*T][MP] (contiguous along m): the
// transpose goes through LDS so that both sides are 256-byte wave rows; other stride patterns read with lanes along m.
__global__ void __launch_bounds__(NTHREADS) weight_pack_kernel(const int* table, int n_entries, const float* src, float* dst, int64_t total) {
    __shared__ float tile[64][65];
    const int64_t e0 = (int64_t)blockIdx.x * 4096;
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)table[mid * 12 + 11] <= e0) lo = mid; else hi = mid - 1;
    }
    const int* r = table + lo * 12;
    const int M = r[2], K = r[3], T = r[4], s_m = r[5], s_k = r[6], s_t = r[7], flip = r[8] & 1, MP = r[9];     // (bit 1 of r[8]: quad copy wanted)
    const int u = (int)((e0 - r[11]) >> 12);
    const int mbs = MP >> 6;
    const int kk0 = (u / mbs) * 64, m0 = (u % mbs) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* sp = src + r[0];
    const bool kfast = (s_t == 1 && s_k == T && !flip);      // (k,tap) contiguous in the source
    if (kfast) {
        const int kk = kk0 + lane;
        for (int mi = wave; mi < 64; mi += 4) {
            const int m = m0 + mi;
            tile[lane][mi] = (m < M && kk < K * T) ? sp[(int64_t)m * s_m + kk] : 0.0f;
        }
    } else {
        const int m = m0 + lane;
        for (int ki = wave; ki < 64; ki += 4) {
            const int kk = kk0 + ki;
            const int kc = kk / T, tap = kk - kc * T;
            tile[ki][lane] = (m < M && kc < K) ? sp[(int64_t)m * s_m + (int64_t)kc * s_k + (flip ? T - 1 - tap : tap) * s_t] : 0.0f;
        }
    }
    __syncthreads();
    float* dp = dst + r[1];
    for (int ki = wave; ki < 64; ki += 4) dp[(int64_t)(kk0 + ki) * MP + m0 + lane] = tile[ki][lane];
}

// bf16-mixed plans: a second copy of every packed entry, rounded to bf16 (RNE) in the fragment order of the bf16 MFMA kernels,
// [KP/8][T][MP][8] (eight consecutive input channels of one (tap, output row) are one 16-byte unit = one lane's MFMA A operand).
// Reads the f32 pack the kernel above has just written (L2-hot, lanes along m), one 16-byte store per thread.
__global__ void __launch_bounds__(NTHREADS) weight_pack_bf16_kernel(const int* table, int n_entries, const float* packed, uint4* dst16) {
    const int64_t e0 = (int64_t)blockIdx.x * (NTHREADS * 8);       // first f32 element of this block's units
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)table[mid * 12 + 11] <= e0) lo = mid; else hi = mid - 1;
    }
    const int* r = table + lo * 12;
    const int T = r[4], MP = r[9];
    const int64_t v = ((e0 - r[11]) >> 3) + threadIdx.x;           // unit inside the entry: (ob * T + tap) * MP + m
    const int m = (int)(v % MP);
    const int64_t ot = v / MP;
    const int tap = (int)(ot % T);
    const int64_t ob = ot / T;
    const float* sp = packed + r[1];
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = sp[((ob * 8 + q) * T + tap) * MP + m];
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint4 w;
    { f2 a = {x[0], x[1]}; w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2)); }
    { f2 a = {x[2], x[3]}; w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2)); }
    { f2 a = {x[4], x[5]}; w.z = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2)); }
    { f2 a = {x[6], x[7]}; w.w = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2)); }
    dst16[(r[1] >> 3) + v] = w;                                   // ushort offset r[1] of the bf16 region = unit r[1] / 8
}

// f32 plans: a second copy of every 1x1 entry (T = 1) in the "quad" layout of csrc/conv_q4.hip, [KP/8][MP][8] with the eight input
// channels of a group in the order (k & 1) * 4 + (k >> 1): the A operands of four consecutive MFMA k-steps (k = 2 s + lane / 32)
// are then 16 contiguous bytes of LDS.  Reads the f32 pack the kernel above has just written (L2-hot, lanes along m), two 16-byte
// stores per thread.
__global__ void __launch_bounds__(NTHREADS) weight_pack_q4_kernel(const int* table, int n_entries, const float* packed, float* mirror) {
    const int64_t e0 = (int64_t)blockIdx.x * (NTHREADS * 8);       // first f32 element of this block's units
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)table[mid * 12 + 11] <= e0) lo = mid; else hi = mid - 1;
    }
    const int* r = table + lo * 12;
    if (r[4] != 1 || !(r[8] & 2)) return;                          // only the 1x1 entries a FLAG_Q4 stage reads (plan: mark_q4)
    const int MP = r[9];
    const int64_t v = ((e0 - r[11]) >> 3) + threadIdx.x;           // unit inside the entry: kg * MP + m
    const int m = (int)(v % MP);
    const int64_t kg = v / MP;
    const float* sp = packed + r[1];
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = sp[(kg * 8 + q) * MP + m];
    f32x4 w0 = {x[0], x[2], x[4], x[6]}, w1 = {x[1], x[3], x[5], x[7]};
    f32x4* dp = reinterpret_cast<f32x4*>(mirror + r[1] + v * 8);
    dp[0] = w0;
    dp[1] = w1;
}

int launch_weight_pack(const S2kOp& op, const Ctx& c) {
    const int* table = ref_ptr<const int>(c, op.t[S2K_WEIGHT_PACK_T_TABLE]);
    const float* src = ref_ptr<const float>(c, op.t[S2K_WEIGHT_PACK_T_SRC]);
    float* dst = ref_ptr<float>(c, op.t[S2K_WEIGHT_PACK_T_DST]);
    CHECK_PTRS("weight_pack", table, src, dst);
    const int64_t total = op.n[S2K_WEIGHT_PACK_N_TOTAL];
    const int n = op.d[S2K_WEIGHT_PACK_D_N_ENTRIES];
    if (!table || !src || !dst || n <= 0 || total <= 0 || (total & 4095) || (total >> 12) > 0x7fffffff) {
        set_error("weight_pack: bad args (total must be a multiple of 4096: MP % 128 == 0, KP % 64 == 0)"); return S2K_EINVAL;
    }
    hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)(total >> 12)), dim3(NTHREADS), 0, c.stream, table, n, src, dst, total);
    const int64_t b16 = op.n[S2K_WEIGHT_PACK_N_BF16_BASE];
    if (b16 > 0) {
        if (b16 & 15) { set_error("weight_pack: BF16_BASE must be a multiple of 16 bytes"); return S2K_EINVAL; }
        hipLaunchKernelGGL(weight_pack_bf16_kernel, dim3((unsigned)(total / (NTHREADS * 8))), dim3(NTHREADS), 0, c.stream, table, n,
                           dst, reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + b16));
    }
    const int64_t q4 = op.n[S2K_WEIGHT_PACK_N_Q4_BASE];
    if (q4 > 0) {
        if (q4 & 15) { set_error("weight_pack: Q4_BASE must be a multiple of 16 bytes"); return S2K_EINVAL; }
        hipLaunchKernelGGL(weight_pack_q4_kernel, dim3((unsigned)(total / (NTHREADS * 8))), dim3(NTHREADS), 0, c.stream, table, n,
                           dst, reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + q4));
    }
    return S2K_OK;
}

// ---------------- BN_FINALIZE ----------------------------------------------------------------------
// One wave per channel: the (up to 64) statistics replicas are read by the lanes in parallel and summed with a wave
// reduction (a thread-per-channel loop over the replicas was 64 dependent L2 round trips: 12 us per launch, 252 launches a step)
__global__ void bn_finalize_kernel(const double* stats, const float* gamma, const float* beta, float* rm, float* rv,
                                   float* bnv, int C, int train, double count, float eps, float mom, int nrep) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    float mean, invstd;
    if (train) {
        double s = 0.0, q = 0.0;
        for (int r = lane; r < nrep; r += 64) { s += stats[(int64_t)r * 2 * C + c]; q += stats[(int64_t)r * 2 * C + C + c]; }
        s = wave_sum_d(s);
        q = wave_sum_d(q);
        const double m = s / count;
        double var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        mean = (float)m;
        const double unbiased = var * (count / fmax(count - 1.0, 1.0));
        if (lane == 0) {
            rm[c] = (1.0f - mom) * rm[c] + mom * mean;
            rv[c] = (1.0f - mom) * rv[c] + mom * (float)unbiased;
        }
    } else {
        mean = rm[c];
        invstd = 1.0f / sqrtf(rv[c] + eps);
    }
    if (lane != 0) return;
    const float scale = gamma[c] * invstd;
    bnv[c] = scale;
    bnv[C + c] = beta[c] - mean * scale;
    bnv[2 * C + c] = mean;
    bnv[3 * C + c] = invstd;
}

int launch_bn_finalize(const S2kOp& op, const Ctx& c) {
    const double* stats = ref_ptr<const double>(c, op.t[S2K_BN_FINALIZE_T_STATS]);
    const float* gamma = ref_ptr<const float>(c, op.t[S2K_BN_FINALIZE_T_GAMMA]);
    const float* beta = ref_ptr<const float>(c, op.t[S2K_BN_FINALIZE_T_BETA]);
    float* rm = ref_ptr<float>(c, op.t[S2K_BN_FINALIZE_T_RM]);
    float* rv = ref_ptr<float>(c, op.t[S2K_BN_FINALIZE_T_RV]);
    float* bnv = ref_ptr<float>(c, op.t[S2K_BN_FINALIZE_T_BNV]);
    CHECK_PTRS("bn_finalize", stats, gamma, beta, rm, rv, bnv);
    const int C = op.d[S2K_BN_FINALIZE_D_C], train = op.d[S2K_BN_FINALIZE_D_TRAIN];
    if (!gamma || !beta || !rm || !rv || !bnv || C <= 0 || (train && !stats)) { set_error("bn_finalize: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, c.stream, stats, gamma, beta, rm, rv, bnv, C,
                       train, (double)op.n[S2K_BN_FINALIZE_N_COUNT], op.f[S2K_BN_FINALIZE_F_EPS], op.f[S2K_BN_FINALIZE_F_MOM],
                       op.d[S2K_BN_FINALIZE_D_NREP] > 0 ? op.d[S2K_BN_FINALIZE_D_NREP] : 1);
    return S2K_OK;
}

// ---------------- plane reductions: SE_POOL, SE_BWD_REDUCE, CHANNEL_SUM ----------------------------------
// MODE 0: out[plane] = sum act(affine(y)) / HW      (SE_POOL; whole plane per wave, CHUNK ignored)
// MODE 1: out[plane] = sum g * act(affine(y))       (SE_BWD_REDUCE)
// MODE 2: out[c]    += sum g                        (CHANNEL_SUM, float atomics)
template <int MODE, bool VEC>
__global__ void __launch_bounds__(NTHREADS) plane_reduce_kernel(const float* g, const float* y, const float* bnv,
                                                                float* out, int C, int HW, int64_t nplanes, int pro,
                                                                const BnFold fold = BnFold{}) {
    const int lane = threadIdx.x & 63;
    int64_t plane;
    int start, count;
    if (MODE == 2) {
        Task t;
        if (!get_task(HW, nplanes, t)) return;
        plane = t.plane; start = t.start; count = t.count;
    } else {
        plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (plane >= nplanes) return;
        start = 0; count = HW;
    }
    const int c = (int)((uint32_t)plane % (uint32_t)C);
    float scale = 1.0f, shift = 0.0f;
    if (MODE == 0 && fold.stats) bn_fold_wave(fold, C, c, plane < C, scale, shift);   // BN_FINALIZE folded in (one wave per plane)
    else if (MODE != 2 && pro != S2K_PRO_NONE) { scale = bnv[c]; shift = bnv[C + c]; }
    const int64_t base = plane * HW + start;
    float s = 0.0f;
    if (VEC) {
        const int n4 = count >> 2;
        for (int i = lane; i < n4; i += 64) {
            float4 a = make_float4(0, 0, 0, 0), b = make_float4(1, 1, 1, 1);
            if (MODE != 0) a = reinterpret_cast<const float4*>(g + base)[i];
            if (MODE != 2) {
                b = reinterpret_cast<const float4*>(y + base)[i];
                b.x = apply_pro(b.x, pro, scale, shift); b.y = apply_pro(b.y, pro, scale, shift);
                b.z = apply_pro(b.z, pro, scale, shift); b.w = apply_pro(b.w, pro, scale, shift);
            }
            if (MODE == 0) s += (b.x + b.y) + (b.z + b.w);
            else s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
    } else {
        for (int i = lane; i < count; i += 64) {
            float a = (MODE != 0) ? g[base + i] : 0.0f;
            float b = (MODE != 2) ? apply_pro(y[base + i], pro, scale, shift) : 1.0f;
            s += (MODE == 0) ? b : a * b;
        }
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (MODE == 0) out[plane] = s / (float)HW;
        else if (MODE == 1) out[plane] = s;
        else atomicAdd(out + c, s);
    }
}

// Large planes of few channels (the 128x128 stages: 24 - 144 channels, 768 - 4608 planes): one WORKGROUP per plane - four waves
// split it and combine through LDS.  One wave per plane left three quarters of the SIMDs without a wave and each wave with
// 64 dependent iterations (40 us for a 50 MB tensor).
__global__ void __launch_bounds__(NTHREADS) se_pool_big_kernel(const float* y, const float* bnv, float* out, int C, int HW, int pro,
                                                               const BnFold fold) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.x;
    const int c = (int)((uint32_t)plane % (uint32_t)C);
    float scale = 1.0f, shift = 0.0f;
    if (fold.stats) bn_fold_wave(fold, C, c, plane < C && wave == 0, scale, shift);
    else if (pro != S2K_PRO_NONE) { scale = bnv[c]; shift = bnv[C + c]; }
    const float4* y4 = reinterpret_cast<const float4*>(y + plane * HW);
    const int n4 = HW >> 2;
    float s0 = 0.0f, s1 = 0.0f;
    int i = threadIdx.x;
    for (; i + NTHREADS < n4; i += 2 * NTHREADS) {
        float4 a = y4[i], b = y4[i + NTHREADS];
        a.x = apply_pro(a.x, pro, scale, shift); a.y = apply_pro(a.y, pro, scale, shift);
        a.z = apply_pro(a.z, pro, scale, shift); a.w = apply_pro(a.w, pro, scale, shift);
        b.x = apply_pro(b.x, pro, scale, shift); b.y = apply_pro(b.y, pro, scale, shift);
        b.z = apply_pro(b.z, pro, scale, shift); b.w = apply_pro(b.w, pro, scale, shift);
        s0 += (a.x + a.y) + (a.z + a.w);
        s1 += (b.x + b.y) + (b.z + b.w);
    }
    for (; i < n4; i += NTHREADS) {
        float4 a = y4[i];
        a.x = apply_pro(a.x, pro, scale, shift); a.y = apply_pro(a.y, pro, scale, shift);
        a.z = apply_pro(a.z, pro, scale, shift); a.w = apply_pro(a.w, pro, scale, shift);
        s0 += (a.x + a.y) + (a.z + a.w);
    }
    const float s = wave_sum(s0 + s1);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[plane] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)HW;
}

// Small planes (HW <= 256, the 16x16 and 8x8 stages where C is in the hundreds or thousands): one wave per CHANNEL walks
// the batch - LPP = pow2ceil(HW / 4) lanes hold one plane as float4s, 64 / LPP planes (samples) per pass.  {scale, shift}
// (from BNV, or from the statistics when BN_FINALIZE is folded in) are formed once per channel instead of once per (b, c)
// plane: with one wave per plane the folded arithmetic cost more than the 64-element plane itself.
__global__ void __launch_bounds__(NTHREADS) se_pool_small_kernel(const float* y, const float* bnv, float* out, int B, int C, int HW,
                                                                 int pro, int lpp, const BnFold fold) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float scale = 1.0f, shift = 0.0f;
    if (fold.stats) bn_fold_wave(fold, C, c, blockIdx.y == 0, scale, shift);
    else if (pro != S2K_PRO_NONE) { scale = bnv[c]; shift = bnv[C + c]; }
    const int pp = 64 / lpp, sub = lane / lpp, li = lane & (lpp - 1);
    const bool in = 4 * li < HW;
    const float inv = 1.0f / (float)HW;
    const int bper = (B + gridDim.y - 1) / gridDim.y, b_lo = blockIdx.y * bper;
    B = min(B, b_lo + bper);
    for (int b0 = b_lo; b0 < B; b0 += pp) {
        const int b = b0 + sub;
        float s = 0.0f;
        if (b < B && in) {
            float4 v = *reinterpret_cast<const float4*>(y + ((int64_t)b * C + c) * HW + 4 * li);
            v.x = apply_pro(v.x, pro, scale, shift); v.y = apply_pro(v.y, pro, scale, shift);
            v.z = apply_pro(v.z, pro, scale, shift); v.w = apply_pro(v.w, pro, scale, shift);
            s = (v.x + v.y) + (v.z + v.w);
        }
        s = group_sum(s, lpp);
        if (li == 0 && b < B) out[(int64_t)b * C + c] = s * inv;
    }
}

template <int MODE>
static int launch_plane_reduce(const float* g, const float* y, const float* bnv, float* out, int B, int C, int HW, int pro,
                               hipStream_t st, const BnFold& fold = BnFold{}) {
    const int64_t nplanes = (int64_t)B * C;
    const unsigned blocks = (MODE == 2) ? task_blocks(HW, nplanes) : (unsigned)cdiv64(nplanes, 4);
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((plane_reduce_kernel<MODE, true>), dim3(blocks), dim3(NTHREADS), 0, st, g, y, bnv, out, C, HW, nplanes, pro, fold);
    else
        hipLaunchKernelGGL((plane_reduce_kernel<MODE, false>), dim3(blocks), dim3(NTHREADS), 0, st, g, y, bnv, out, C, HW, nplanes, pro, fold);
    return S2K_OK;
}

int launch_se_pool(const S2kOp& op, const Ctx& c) {
    const float* y = ref_ptr<const float>(c, op.t[S2K_SE_POOL_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_SE_POOL_T_BNV]);
    float* pool = ref_ptr<float>(c, op.t[S2K_SE_POOL_T_POOL]);
    CHECK_PTRS("se_pool", y, bnv, pool);
    const int pro = op.d[S2K_SE_POOL_D_PRO];
    if (!y || !pool || (pro && !bnv)) { set_error("se_pool: bad args"); return S2K_EINVAL; }
    BnFold fold;
    if (int e = fill_bn_fold(fold, c, &op.t[S2K_SE_POOL_T_FSTATS], op.n[S2K_SE_POOL_N_FCOUNT], op.d[S2K_SE_POOL_D_FNREP], op.f[S2K_SE_POOL_F_FEPS],
                             op.f[S2K_SE_POOL_F_FMOM], const_cast<float*>(bnv), "se_pool")) return e;
    if (fold.stats && !pro) { set_error("se_pool: FSTATS without a prologue"); return S2K_EINVAL; }
    {
        const int B = op.d[S2K_SE_POOL_D_B], C = op.d[S2K_SE_POOL_D_C], HW = op.d[S2K_SE_POOL_D_HW];
        if (HW >= 4096 && (HW & 3) == 0 && (int64_t)B * C <= 8192) {
            hipLaunchKernelGGL(se_pool_big_kernel, dim3((unsigned)(B * C)), dim3(NTHREADS), 0, c.stream, y, bnv, pool, C, HW, pro, fold);
            return S2K_OK;
        }
        if (HW <= 256 && (HW & 3) == 0 && B > 1) {
            int lpp = 1;
            while (4 * lpp < HW) lpp <<= 1;
            const int bsplit = std::max(1, std::min(cdiv(B, 64 / lpp), cdiv(2048, C)));     // >= ~2048 waves in all
            hipLaunchKernelGGL(se_pool_small_kernel, dim3(cdiv(C, 4), bsplit), dim3(NTHREADS), 0, c.stream, y, bnv, pool, B, C, HW, pro, lpp, fold);
            return S2K_OK;
        }
    }
    return launch_plane_reduce<0>(nullptr, y, bnv, pool, op.d[S2K_SE_POOL_D_B], op.d[S2K_SE_POOL_D_C], op.d[S2K_SE_POOL_D_HW], pro, c.stream, fold);
}

int launch_se_bwd_reduce(const S2kOp& op, const Ctx& c) {
    const float* g = ref_ptr<const float>(c, op.t[S2K_SE_BWD_REDUCE_T_G]);
    const float* y = ref_ptr<const float>(c, op.t[S2K_SE_BWD_REDUCE_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_SE_BWD_REDUCE_T_BNV]);
    float* out = ref_ptr<float>(c, op.t[S2K_SE_BWD_REDUCE_T_DGATE]);
    CHECK_PTRS("se_bwd_reduce", g, y, bnv, out);
    const int pro = op.d[S2K_SE_BWD_REDUCE_D_PRO];
    if (!g || !y || !out || (pro && !bnv)) { set_error("se_bwd_reduce: bad args"); return S2K_EINVAL; }
    return launch_plane_reduce<1>(g, y, bnv, out, op.d[S2K_SE_BWD_REDUCE_D_B], op.d[S2K_SE_BWD_REDUCE_D_C],
                                  op.d[S2K_SE_BWD_REDUCE_D_HW], pro, c.stream);
}

// ---------------- SE + BatchNorm backward of the depthwise output in two passes over the expanded tensor -------------
// The chain SE_BWD_REDUCE -> SE_FC_BWD -> BN_BWD_REDUCE -> BN_BWD_APPLY reads the (b, c, hw) gradient d and the raw conv
// output y three times and writes twice, because BN_BWD_REDUCE needs the SE result (ADDBC) of the first pass.  Its sums are
// linear in the per-plane factors, g' = (d*mul + add) * a'(u):
//   sum g'      = mul * sum d a'      + add * sum a'
//   sum g' xhat = mul * sum d a' xhat + add * sum a' xhat
// so ONE pass collects dgate = sum d*a and the four plane sums (SE_BN_SUMS), a per-channel kernel combines them once mul /
// add are known (SE_BN_COMBINE), and BN_BWD_APPLY recomputes g' on the fly: 4 reads + 1 write instead of 6 + 2.
template <bool VEC>
__global__ void __launch_bounds__(NTHREADS) se_bn_sums_kernel(const float* g, const float* y, const float* bnv, float* dgate,
                                                              float* ps, int C, int HW, int64_t nplanes) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= nplanes) return;
    const int c = (int)((uint32_t)plane % (uint32_t)C);
    const float scale = bnv[c], shift = bnv[C + c], mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
    const int64_t base = plane * HW;
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, p4 = 0.0f;
    auto acc = [&](float d, float yy) {
        const float u = fmaf(yy, scale, shift);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        const float ap = sg * (1.0f + u * (1.0f - sg)), xh = (yy - mean) * invstd;
        const float t = d * ap;
        p0 = fmaf(d, u * sg, p0);
        p1 += t;
        p2 += ap;
        p3 = fmaf(t, xh, p3);
        p4 = fmaf(ap, xh, p4);
    };
    if (VEC) {
        for (int i = lane; i < (HW >> 2); i += 64) {
            const float4 dv = reinterpret_cast<const float4*>(g + base)[i];
            const float4 yv = reinterpret_cast<const float4*>(y + base)[i];
            acc(dv.x, yv.x); acc(dv.y, yv.y); acc(dv.z, yv.z); acc(dv.w, yv.w);
        }
    } else {
        for (int i = lane; i < HW; i += 64) acc(g[base + i], y[base + i]);
    }
    p0 = wave_sum_hi(p0); p1 = wave_sum_hi(p1); p2 = wave_sum_hi(p2); p3 = wave_sum_hi(p3); p4 = wave_sum_hi(p4);
    if (lane == 63) {
        dgate[plane] = p0;
        ps[plane] = p1;
        ps[nplanes + plane] = p2;
        ps[2 * nplanes + plane] = p3;
        ps[3 * nplanes + plane] = p4;
    }
}

// SE_BN_SUMS on large planes of few channels: one workgroup per plane (see se_pool_big_kernel)
__global__ void __launch_bounds__(NTHREADS) se_bn_sums_big_kernel(const float* g, const float* y, const float* bnv, float* dgate, float* ps,
                                                                  int C, int HW, int64_t nplanes) {
    __shared__ float red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.x;
    const int c = (int)((uint32_t)plane % (uint32_t)C);
    const float scale = bnv[c], shift = bnv[C + c], mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, p4 = 0.0f;
    auto acc = [&](float d, float yy) {
        const float u = fmaf(yy, scale, shift);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        const float ap = sg * (1.0f + u * (1.0f - sg)), xh = (yy - mean) * invstd;
        const float t = d * ap;
        p0 = fmaf(d, u * sg, p0);
        p1 += t;
        p2 += ap;
        p3 = fmaf(t, xh, p3);
        p4 = fmaf(ap, xh, p4);
    };
    const float4* g4 = reinterpret_cast<const float4*>(g + plane * HW);
    const float4* y4 = reinterpret_cast<const float4*>(y + plane * HW);
    const int n4 = HW >> 2;
    int i = threadIdx.x;
    for (; i + NTHREADS < n4; i += 2 * NTHREADS) {
        const float4 d0 = g4[i], y0 = y4[i], d1 = g4[i + NTHREADS], y1 = y4[i + NTHREADS];
        acc(d0.x, y0.x); acc(d0.y, y0.y); acc(d0.z, y0.z); acc(d0.w, y0.w);
        acc(d1.x, y1.x); acc(d1.y, y1.y); acc(d1.z, y1.z); acc(d1.w, y1.w);
    }
    for (; i < n4; i += NTHREADS) {
        const float4 d0 = g4[i], y0 = y4[i];
        acc(d0.x, y0.x); acc(d0.y, y0.y); acc(d0.z, y0.z); acc(d0.w, y0.w);
    }
    p0 = wave_sum_hi(p0); p1 = wave_sum_hi(p1); p2 = wave_sum_hi(p2); p3 = wave_sum_hi(p3); p4 = wave_sum_hi(p4);
    if (lane == 63) { red[wave][0] = p0; red[wave][1] = p1; red[wave][2] = p2; red[wave][3] = p3; red[wave][4] = p4; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (k == 0) dgate[plane] = v;
        else ps[(int64_t)(k - 1) * nplanes + plane] = v;
    }
}

// SE_BN_SUMS on small planes (<= 64 elements): one wave per channel (and batch chunk), 16 lanes per plane, four planes per pass,
// the loads of four passes in flight; the five plane sums are group reductions (one per four planes instead of one per plane)
__global__ void __launch_bounds__(NTHREADS) se_bn_sums_small_kernel(const float* g, const float* y, const float* bnv, float* dgate,
                                                                    float* ps, int B, int C, int HW, int lpp) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float scale = bnv[c], shift = bnv[C + c], mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
    const int pp = 64 / lpp, sub = lane / lpp, li = lane & (lpp - 1);
    const bool in = 4 * li < HW;
    const int64_t nplanes = (int64_t)B * C;
    const int bper = (B + gridDim.y - 1) / gridDim.y, b_lo = blockIdx.y * bper, b_hi = min(B, b_lo + bper);
    constexpr int U = 4;
    for (int b0 = b_lo + sub; b0 - sub < b_hi; b0 += pp * U) {
        float4 dv[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = min(b0 + u * pp, b_hi - 1);
            const int64_t off = ((int64_t)b * C + c) * HW + 4 * (in ? li : 0);
            dv[u] = *reinterpret_cast<const float4*>(g + off);
            yv[u] = *reinterpret_cast<const float4*>(y + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * pp;
            if (b - sub >= b_hi) break;                                   // wave-uniform: the whole pass is past the chunk
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, p4 = 0.0f;
            if (in && b < b_hi) {
                const float dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w}, yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float uu = fmaf(yy[k], scale, shift);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-uu));
                    const float ap = sg * (1.0f + uu * (1.0f - sg)), xh = (yy[k] - mean) * invstd;
                    const float t = dd[k] * ap;
                    p0 = fmaf(dd[k], uu * sg, p0);
                    p1 += t;
                    p2 += ap;
                    p3 = fmaf(t, xh, p3);
                    p4 = fmaf(ap, xh, p4);
                }
            }
            p0 = group_sum(p0, lpp); p1 = group_sum(p1, lpp); p2 = group_sum(p2, lpp); p3 = group_sum(p3, lpp); p4 = group_sum(p4, lpp);
            if (li == 0 && b < b_hi) {
                const int64_t plane = (int64_t)b * C + c;
                dgate[plane] = p0;
                ps[plane] = p1;
                ps[nplanes + plane] = p2;
                ps[2 * nplanes + plane] = p3;
                ps[3 * nplanes + plane] = p4;
            }
        }
    }
}

int launch_se_bn_sums(const S2kOp& op, const Ctx& c) {
    const float* g = ref_ptr<const float>(c, op.t[S2K_SE_BN_SUMS_T_G]);
    const float* y = ref_ptr<const float>(c, op.t[S2K_SE_BN_SUMS_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_SE_BN_SUMS_T_BNV]);
    float* dgate = ref_ptr<float>(c, op.t[S2K_SE_BN_SUMS_T_DGATE]);
    float* ps = ref_ptr<float>(c, op.t[S2K_SE_BN_SUMS_T_PS]);
    CHECK_PTRS("se_bn_sums", g, y, bnv, dgate, ps);
    const int B = op.d[S2K_SE_BN_SUMS_D_B], C = op.d[S2K_SE_BN_SUMS_D_C], HW = op.d[S2K_SE_BN_SUMS_D_HW];
    if (!g || !y || !bnv || !dgate || !ps || B <= 0 || C <= 0 || HW <= 0 || op.d[S2K_SE_BN_SUMS_D_ACT] != S2K_PRO_SILU) {
        set_error("se_bn_sums: bad args (SiLU only)"); return S2K_EINVAL;
    }
    const int64_t nplanes = (int64_t)B * C;
    if (HW >= 4096 && (HW & 3) == 0 && nplanes <= 8192) {
        hipLaunchKernelGGL(se_bn_sums_big_kernel, dim3((unsigned)nplanes), dim3(NTHREADS), 0, c.stream, g, y, bnv, dgate, ps, C, HW, nplanes);
        return S2K_OK;
    }
    if (HW <= 64 && (HW & 3) == 0 && B > 1) {
        int lpp = 1;
        while (4 * lpp < HW) lpp <<= 1;
        const int bsplit = std::max(1, std::min(cdiv(B, 64 / lpp), cdiv(2048, C)));
        hipLaunchKernelGGL(se_bn_sums_small_kernel, dim3(cdiv(C, 4), bsplit), dim3(NTHREADS), 0, c.stream, g, y, bnv, dgate, ps, B, C, HW, lpp);
        return S2K_OK;
    }
    const unsigned blocks = (unsigned)cdiv64(nplanes, 4);
    if ((HW & 3) == 0) hipLaunchKernelGGL((se_bn_sums_kernel<true>), dim3(blocks), dim3(NTHREADS), 0, c.stream, g, y, bnv, dgate, ps, C, HW, nplanes);
    else hipLaunchKernelGGL((se_bn_sums_kernel<false>), dim3(blocks), dim3(NTHREADS), 0, c.stream, g, y, bnv, dgate, ps, C, HW, nplanes);
    return S2K_OK;
}

// one wave per channel, lanes over the batch (a thread-per-channel loop over b was 32 dependent round trips: 29 us a launch)
__global__ void __launch_bounds__(NTHREADS) se_bn_combine_kernel(const float* ps, const float* mulbc, const float* addbc, double* st2,
                                                                 int B, int C, float addscale) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const int64_t np = (int64_t)B * C;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < B; b += 64) {
        const int64_t pl = (int64_t)b * C + c;
        const double mul = mulbc ? (double)mulbc[pl] : 1.0, add = addbc ? (double)addbc[pl] * (double)addscale : 0.0;
        s1 += mul * (double)ps[pl] + add * (double)ps[np + pl];
        s2 += mul * (double)ps[2 * np + pl] + add * (double)ps[3 * np + pl];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane == 0) {
        st2[c] = s1;
        st2[C + c] = s2;
    }
}

int launch_se_bn_combine(const S2kOp& op, const Ctx& c) {
    const float* ps = ref_ptr<const float>(c, op.t[S2K_SE_BN_COMBINE_T_PS]);
    const float* mulbc = ref_ptr<const float>(c, op.t[S2K_SE_BN_COMBINE_T_MULBC]);
    const float* addbc = ref_ptr<const float>(c, op.t[S2K_SE_BN_COMBINE_T_ADDBC]);
    double* st2 = ref_ptr<double>(c, op.t[S2K_SE_BN_COMBINE_T_STATS2]);
    CHECK_PTRS("se_bn_combine", ps, mulbc, addbc, st2);
    const int B = op.d[S2K_SE_BN_COMBINE_D_B], C = op.d[S2K_SE_BN_COMBINE_D_C];
    if (!ps || !st2 || B <= 0 || C <= 0) { set_error("se_bn_combine: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(se_bn_combine_kernel, dim3(cdiv(C, 4)), dim3(NTHREADS), 0, c.stream, ps, mulbc, addbc, st2, B, C, op.f[S2K_SE_BN_COMBINE_F_ADDSCALE]);
    return S2K_OK;
}

// short planes (token axes of the ViT: 50 .. 197 elements): one workgroup owns a channel, its four waves walk the batch
// interleaved with 8 independent row loads in flight each, and ONE lane adds the result - no atomics (a wave per (b, c)
// plane would be one load and one atomic each; one wave per channel - the first version - was a chain of B/4 dependent
// round trips: 27 us for a 10 MB tensor)
__global__ void __launch_bounds__(NTHREADS) channel_sum_rows_kernel(const float* g, float* out, int B, int C, int HW) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const int64_t bs = (int64_t)C * HW;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    int b = wave;
    for (; b + 28 < B; b += 32) {                   // rows b, b + 4, ..., b + 28 of this wave
        const float* r0 = g + ((int64_t)b * C + c) * HW;
        for (int i = lane; i < HW; i += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += r0[4 * u * bs + i];
        }
    }
    for (; b < B; b += 4) {
        const float* r0 = g + ((int64_t)b * C + c) * HW;
        for (int i = lane; i < HW; i += 64) acc[0] += r0[i];
    }
    const float s = wave_sum_hi(((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
    if (lane == 63) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] += (red[0] + red[1]) + (red[2] + red[3]);
}

int launch_channel_sum(const S2kOp& op, const Ctx& c) {
    const float* g = ref_ptr<const float>(c, op.t[S2K_CHANNEL_SUM_T_G]);
    float* out = ref_ptr<float>(c, op.t[S2K_CHANNEL_SUM_T_OUT]);
    CHECK_PTRS("channel_sum", g, out);
    if (!g || !out) { set_error("channel_sum: bad args"); return S2K_EINVAL; }
    const int B = op.d[S2K_CHANNEL_SUM_D_B], C = op.d[S2K_CHANNEL_SUM_D_C], HW = op.d[S2K_CHANNEL_SUM_D_HW];
    if (B > 0 && C >= 256 && HW > 0 && HW <= 512) {
        hipLaunchKernelGGL(channel_sum_rows_kernel, dim3(C), dim3(NTHREADS), 0, c.stream, g, out, B, C, HW);
        return S2K_OK;
    }
    return launch_plane_reduce<2>(g, nullptr, nullptr, out, op.d[S2K_CHANNEL_SUM_D_B], op.d[S2K_CHANNEL_SUM_D_C],
                                  op.d[S2K_CHANNEL_SUM_D_HW], 0, c.stream);
}

// ---------------- SE FCs ---------------------------------------------------------------------------------
// Tiny GEMMs (B x C x C/24): parallelised over (sample, output) so that a few hundred workgroups exist
// instead of one per sample.
// hpre[b][j] = b1[j] + sum_c w1[j][c] * pool[b][c]            one wave per (b, j), lanes over c
__global__ void __launch_bounds__(NTHREADS) se_fc1_kernel(const float* pool, const float* w1, const float* b1, float* hpre,
                                                          int B, int C, int Q) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (j >= Q) return;
    const float* wr = w1 + (int64_t)j * C;
    const float* pr = pool + (int64_t)b * C;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;      // eight loads in flight per lane, four independent chains
    int i = lane;
    for (; i + 192 < C; i += 256) {
        s0 = fmaf(wr[i], pr[i], s0);
        s1 = fmaf(wr[i + 64], pr[i + 64], s1);
        s2 = fmaf(wr[i + 128], pr[i + 128], s2);
        s3 = fmaf(wr[i + 192], pr[i + 192], s3);
    }
    for (; i < C; i += 64) s0 = fmaf(wr[i], pr[i], s0);
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) hpre[(int64_t)b * Q + j] = s + b1[j];
}

// gate[b][c] = sigmoid(b2[c] + sum_j w2[c][j] * silu(hpre[b][j]))     one thread per (b, c)
__global__ void __launch_bounds__(NTHREADS) se_fc2_kernel(const float* hpre, const float* w2, const float* b2, float* gate,
                                                          int B, int C, int Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < Q; j += NTHREADS) smem[j] = silu_f(hpre[(int64_t)b * Q + j]);
    __syncthreads();
    const int c = blockIdx.x * NTHREADS + threadIdx.x;
    if (c >= C) return;
    const float* wr = w2 + (int64_t)c * Q;
    float s = b2[c];
    int j = 0;
    for (; j + 4 <= Q; j += 4)
        s += (wr[j] * smem[j] + wr[j + 1] * smem[j + 1]) + (wr[j + 2] * smem[j + 2] + wr[j + 3] * smem[j + 3]);
    for (; j < Q; ++j) s = fmaf(wr[j], smem[j], s);
    gate[(int64_t)b * C + c] = 1.0f / (1.0f + __expf(-s));
}

int launch_se_fc(const S2kOp& op, const Ctx& c) {
    const float* pool = ref_ptr<const float>(c, op.t[S2K_SE_FC_T_POOL]);
    const float* w1 = ref_ptr<const float>(c, op.t[S2K_SE_FC_T_W1]);
    const float* b1 = ref_ptr<const float>(c, op.t[S2K_SE_FC_T_B1]);
    const float* w2 = ref_ptr<const float>(c, op.t[S2K_SE_FC_T_W2]);
    const float* b2 = ref_ptr<const float>(c, op.t[S2K_SE_FC_T_B2]);
    float* hpre = ref_ptr<float>(c, op.t[S2K_SE_FC_T_HPRE]);
    float* gate = ref_ptr<float>(c, op.t[S2K_SE_FC_T_GATE]);
    CHECK_PTRS("se_fc", pool, w1, b1, w2, b2, hpre, gate);
    const int B = op.d[S2K_SE_FC_D_B], C = op.d[S2K_SE_FC_D_C], Q = op.d[S2K_SE_FC_D_CSQ];
    if (!pool || !w1 || !b1 || !w2 || !b2 || !hpre || !gate || B <= 0 || C <= 0 || Q <= 0 || Q > 8192 || B > 65535) {
        set_error("se_fc: bad args"); return S2K_EINVAL;
    }
    hipLaunchKernelGGL(se_fc1_kernel, dim3(cdiv(Q, 4), B), dim3(NTHREADS), 0, c.stream, pool, w1, b1, hpre, B, C, Q);
    hipLaunchKernelGGL(se_fc2_kernel, dim3(cdiv(C, NTHREADS), B), dim3(NTHREADS), Q * sizeof(float), c.stream, hpre, w2, b2, gate, B, C, Q);
    return S2K_OK;
}

// backward, phase A1: dgp = dgate*gate*(1-gate); dh[b][j] = sum_c w2[c][j]*dgp[b][c];
// hs = silu(hpre); dhp = dh*silu'(hpre) (overwrites hpre).  One workgroup (16 waves) per (64 j's, sample):
// dgp of the sample staged in LDS, lanes over j (coalesced rows of w2), waves split the channels with 4
// independent accumulators, partials combined through LDS.
__global__ void __launch_bounds__(1024) se_fc_bwd_a1_kernel(const float* dgate, const float* gate, float* hpre, const float* w2,
                                                            float* hs, int B, int C, int Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sdg = smem;              // [C]
    float* part = smem + C;         // [16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 1024) {
        const float g = gate[(int64_t)b * C + i];
        sdg[i] = dgate[(int64_t)b * C + i] * g * (1.0f - g);
    }
    __syncthreads();
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (j < Q) {
        int i = wave;
        for (; i + 112 < C; i += 128) {              // eight row loads in flight per lane (a 3072-channel layer was 48 dependent trips)
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = w2[(int64_t)(i + 16 * u) * Q + j];
            s0 = fmaf(w[0], sdg[i], s0);
            s1 = fmaf(w[1], sdg[i + 16], s1);
            s2 = fmaf(w[2], sdg[i + 32], s2);
            s3 = fmaf(w[3], sdg[i + 48], s3);
            s0 = fmaf(w[4], sdg[i + 64], s0);
            s1 = fmaf(w[5], sdg[i + 80], s1);
            s2 = fmaf(w[6], sdg[i + 96], s2);
            s3 = fmaf(w[7], sdg[i + 112], s3);
        }
        for (; i < C; i += 16) s0 = fmaf(w2[(int64_t)i * Q + j], sdg[i], s0);
    }
    part[wave * 64 + lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && j < Q) {
        float dh = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += part[w * 64 + lane];
        const float hp = hpre[(int64_t)b * Q + j];
        hs[(int64_t)b * Q + j] = silu_f(hp);
        hpre[(int64_t)b * Q + j] = dh * act_grad(hp, S2K_PRO_SILU);   // hpre now holds dhp
    }
}

// phase A2: dgate <- dgp (in place), dpool[b][c] = sum_j w1[j][c] * dhp[b][j]      one thread per (b, c)
__global__ void __launch_bounds__(NTHREADS) se_fc_bwd_a2_kernel(float* dgate, const float* gate, const float* dhp, const float* w1,
                                                                float* dpool, int B, int C, int Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < Q; j += NTHREADS) smem[j] = dhp[(int64_t)b * Q + j];
    __syncthreads();
    const int c = blockIdx.x * NTHREADS + threadIdx.x;
    if (c >= C) return;
    const float g = gate[(int64_t)b * C + c];
    dgate[(int64_t)b * C + c] *= g * (1.0f - g);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int j = 0;
    for (; j + 8 <= Q; j += 8) {                     // eight row loads in flight per lane
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = w1[(int64_t)(j + u) * C + c];
        s0 = fmaf(w[0], smem[j], s0);
        s1 = fmaf(w[1], smem[j + 1], s1);
        s2 = fmaf(w[2], smem[j + 2], s2);
        s3 = fmaf(w[3], smem[j + 3], s3);
        s0 = fmaf(w[4], smem[j + 4], s0);
        s1 = fmaf(w[5], smem[j + 5], s1);
        s2 = fmaf(w[6], smem[j + 6], s2);
        s3 = fmaf(w[7], smem[j + 7], s3);
    }
    for (; j + 4 <= Q; j += 4) {
        s0 = fmaf(w1[(int64_t)j * C + c], smem[j], s0);
        s1 = fmaf(w1[(int64_t)(j + 1) * C + c], smem[j + 1], s1);
        s2 = fmaf(w1[(int64_t)(j + 2) * C + c], smem[j + 2], s2);
        s3 = fmaf(w1[(int64_t)(j + 3) * C + c], smem[j + 3], s3);
    }
    for (; j < Q; ++j) s0 = fmaf(w1[(int64_t)j * C + c], smem[j], s0);
    dpool[(int64_t)b * C + c] = (s0 + s1) + (s2 + s3);
}

// backward, phase B: parameter gradients, summed over the batch inside the thread (no atomics)
__global__ void __launch_bounds__(NTHREADS) se_fc_bwd_b_kernel(const float* dgp, const float* hs, const float* dhp,
                                                               const float* pool, float* dw1, float* db1, float* dw2,
                                                               float* db2, int B, int C, int Q) {
    const int64_t total = (int64_t)C * Q;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (C * Q fits 32 bits - the launchers cap Q at 8192 and C at ~16 k - and a 64-bit division per element cost more than the sum)
    for (int64_t e = t0; e < total; e += stride) {  // dw2[c][j], j fastest
        const int cc = (int)((uint32_t)e / (uint32_t)Q), j = (int)(e - (int64_t)cc * Q);
        float s = 0.0f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s = fmaf(dgp[(int64_t)b * C + cc], hs[(int64_t)b * Q + j], s);
        dw2[e] += s;
    }
    for (int64_t e = t0; e < total; e += stride) {  // dw1[j][c], c fastest
        const int j = (int)((uint32_t)e / (uint32_t)C), cc = (int)(e - (int64_t)j * C);
        float s = 0.0f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s = fmaf(dhp[(int64_t)b * Q + j], pool[(int64_t)b * C + cc], s);
        dw1[e] += s;
    }
    for (int64_t e = t0; e < C; e += stride) {
        float s = 0.0f;
        for (int b = 0; b < B; ++b) s += dgp[(int64_t)b * C + e];
        db2[e] += s;
    }
    for (int64_t e = t0; e < Q; e += stride) {
        float s = 0.0f;
        for (int b = 0; b < B; ++b) s += dhp[(int64_t)b * Q + e];
        db1[e] += s;
    }
}

// The same parameter gradients from LDS (Q <= SEW_QMAX, every EfficientNet): the four operands are a few hundred KB, so the kernel above is a
// chain of dependent L2 round trips (32 x 2 loads per output, four loops: 18 us for microseconds of arithmetic).  Here a workgroup owns
// SEW_CT channels: one round trip brings 32 samples of hs / dhp (whole) and of its dgp / pool columns into LDS, the sums run from
// there in the same sample order (bit-identical results), and the tiles are added to dw2[c][.] / dw1[.][c] once at the end.
constexpr int SEW_CT = 8, SEW_BT = 32, SEW_QMAX = 224, SEW_U = SEW_CT * SEW_QMAX / NTHREADS;
__global__ void __launch_bounds__(NTHREADS) se_fc_wgrad_tile_kernel(const float* dgp, const float* hs, const float* dhp, const float* pool,
                                                                    float* dw1, float* db1, float* dw2, float* db2, int B, int C, int Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs_s = smem;                         // [SEW_BT][Q]
    float* dhp_s = hs_s + SEW_BT * Q;           // [SEW_BT][Q]
    float* dgp_s = dhp_s + SEW_BT * Q;          // [SEW_BT][SEW_CT]
    float* pool_s = dgp_s + SEW_BT * SEW_CT;    // [SEW_BT][SEW_CT]
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * SEW_CT, nc = min(SEW_CT, C - c0);
    const int per = SEW_CT * Q;                 // elements of each of the two tiles, <= SEW_U per thread
    float a1[SEW_U], a2[SEW_U];
#pragma unroll
    for (int u = 0; u < SEW_U; ++u) a1[u] = a2[u] = 0.0f;
    float sb1 = 0.0f, sb2 = 0.0f;
    for (int b0 = 0; b0 < B; b0 += SEW_BT) {
        const int nb = min(SEW_BT, B - b0);
        if (b0) __syncthreads();
        for (int i = t; i < nb * Q; i += NTHREADS) {          // rows b0 .. b0 + nb of [B][Q] are one contiguous range
            hs_s[i] = hs[(int64_t)b0 * Q + i];
            dhp_s[i] = dhp[(int64_t)b0 * Q + i];
        }
        for (int i = t; i < nb * SEW_CT; i += NTHREADS) {
            const int b = i / SEW_CT, k = i % SEW_CT;
            const bool in = k < nc;
            dgp_s[i] = in ? dgp[(int64_t)(b0 + b) * C + c0 + k] : 0.0f;
            pool_s[i] = in ? pool[(int64_t)(b0 + b) * C + c0 + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SEW_U; ++u) {
            const int e = t + u * NTHREADS;
            if (e < per) {
                const int c2 = (int)((uint32_t)e / (uint32_t)Q), j2 = e - c2 * Q;       // dw2 tile [SEW_CT][Q], j fastest
                const int j1 = e / SEW_CT, c1 = e % SEW_CT;                             // dw1 tile [Q][SEW_CT], c fastest
                float s1 = a1[u], s2 = a2[u];
                for (int b = 0; b < nb; ++b) {
                    s2 = fmaf(dgp_s[b * SEW_CT + c2], hs_s[b * Q + j2], s2);
                    s1 = fmaf(dhp_s[b * Q + j1], pool_s[b * SEW_CT + c1], s1);
                }
                a1[u] = s1; a2[u] = s2;
            }
        }
        if (t < SEW_CT) for (int b = 0; b < nb; ++b) sb2 += dgp_s[b * SEW_CT + t];
        if (blockIdx.x == 0 && t < Q) for (int b = 0; b < nb; ++b) sb1 += dhp_s[b * Q + t];
    }
#pragma unroll
    for (int u = 0; u < SEW_U; ++u) {
        const int e = t + u * NTHREADS;
        if (e < per) {
            const int c2 = (int)((uint32_t)e / (uint32_t)Q), j2 = e - c2 * Q;
            const int j1 = e / SEW_CT, c1 = e % SEW_CT;
            if (c2 < nc) dw2[(int64_t)(c0 + c2) * Q + j2] += a2[u];
            if (c1 < nc) dw1[(int64_t)j1 * C + c0 + c1] += a1[u];
        }
    }
    if (t < nc) db2[c0 + t] += sb2;
    if (blockIdx.x == 0 && t < Q) db1[t] += sb1;
}

static void launch_se_fc_param_grads(const float* dgp, const float* hs, const float* dhp, const float* pool, float* dw1, float* db1, float* dw2,
                                     float* db2, int B, int C, int Q, hipStream_t st) {
    if (Q <= SEW_QMAX) {
        const size_t lds = (size_t)(2 * SEW_BT * Q + 2 * SEW_BT * SEW_CT) * sizeof(float);       // <= 59,392 bytes
        hipLaunchKernelGGL(se_fc_wgrad_tile_kernel, dim3(cdiv(C, SEW_CT)), dim3(NTHREADS), lds, st, dgp, hs, dhp, pool, dw1, db1, dw2, db2, B, C, Q);
        return;
    }
    const int blocks = (int)std::min<int64_t>(cdiv64((int64_t)C * Q, 256), 1024);
    hipLaunchKernelGGL(se_fc_bwd_b_kernel, dim3(blocks), dim3(NTHREADS), 0, st, dgp, hs, dhp, pool, dw1, db1, dw2, db2, B, C, Q);
}

// Phases A1 + A2 in one launch for the layers whose squeeze width fits a wave (Q <= 64, 28 of the 39 blocks of a b5): one
// workgroup of 16 waves per sample; the two-kernel form costs two launches for microseconds of arithmetic.  (The same merge of
// the FORWARD pair was slower: its first Linear wants Q / 4 workgroups per sample, not one.)
__global__ void __launch_bounds__(1024) se_fc_bwd_small_kernel(float* dgate, const float* gate, float* hpre, const float* w1, const float* w2,
                                                               float* hs, float* dpool, int C, int Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sdg = smem;              // [C]
    float* part = smem + C;         // [16][64]
    float* dhs = part + 16 * 64;    // [64] dhp of this sample
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < C; i += 1024) {
        const float g = gate[(int64_t)b * C + i];
        const float v = dgate[(int64_t)b * C + i] * g * (1.0f - g);
        sdg[i] = v;
        dgate[(int64_t)b * C + i] = v;                  // DGATE <- d(pre-sigmoid), as phase A2 leaves it
    }
    __syncthreads();
    const int j = lane;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (j < Q) {
        int i = wave;
        for (; i + 112 < C; i += 128) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = w2[(int64_t)(i + 16 * u) * Q + j];
            s0 = fmaf(w[0], sdg[i], s0);
            s1 = fmaf(w[1], sdg[i + 16], s1);
            s2 = fmaf(w[2], sdg[i + 32], s2);
            s3 = fmaf(w[3], sdg[i + 48], s3);
            s0 = fmaf(w[4], sdg[i + 64], s0);
            s1 = fmaf(w[5], sdg[i + 80], s1);
            s2 = fmaf(w[6], sdg[i + 96], s2);
            s3 = fmaf(w[7], sdg[i + 112], s3);
        }
        for (; i < C; i += 16) s0 = fmaf(w2[(int64_t)i * Q + j], sdg[i], s0);
    }
    part[wave * 64 + lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0) {
        float dhp = 0.0f;
        if (j < Q) {
            float dh = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) dh += part[w * 64 + lane];
            const float hp = hpre[(int64_t)b * Q + j];
            hs[(int64_t)b * Q + j] = silu_f(hp);
            dhp = dh * act_grad(hp, S2K_PRO_SILU);
            hpre[(int64_t)b * Q + j] = dhp;             // HPRE now holds dhp
        }
        dhs[lane] = dhp;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        int jj = 0;
        for (; jj + 8 <= Q; jj += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = w1[(int64_t)(jj + u) * C + c];
            t0 = fmaf(w[0], dhs[jj], t0);
            t1 = fmaf(w[1], dhs[jj + 1], t1);
            t2 = fmaf(w[2], dhs[jj + 2], t2);
            t3 = fmaf(w[3], dhs[jj + 3], t3);
            t0 = fmaf(w[4], dhs[jj + 4], t0);
            t1 = fmaf(w[5], dhs[jj + 5], t1);
            t2 = fmaf(w[6], dhs[jj + 6], t2);
            t3 = fmaf(w[7], dhs[jj + 7], t3);
        }
        for (; jj + 4 <= Q; jj += 4) {
            t0 = fmaf(w1[(int64_t)jj * C + c], dhs[jj], t0);
            t1 = fmaf(w1[(int64_t)(jj + 1) * C + c], dhs[jj + 1], t1);
            t2 = fmaf(w1[(int64_t)(jj + 2) * C + c], dhs[jj + 2], t2);
            t3 = fmaf(w1[(int64_t)(jj + 3) * C + c], dhs[jj + 3], t3);
        }
        for (; jj < Q; ++jj) t0 = fmaf(w1[(int64_t)jj * C + c], dhs[jj], t0);
        dpool[(int64_t)b * C + c] = (t0 + t1) + (t2 + t3);
    }
}

int launch_se_fc_bwd(const S2kOp& op, const Ctx& c) {
    float* dgate = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DGATE]);
    const float* gate = ref_ptr<const float>(c, op.t[S2K_SE_FC_BWD_T_GATE]);
    float* hpre = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_HPRE]);
    const float* pool = ref_ptr<const float>(c, op.t[S2K_SE_FC_BWD_T_POOL]);
    const float* w1 = ref_ptr<const float>(c, op.t[S2K_SE_FC_BWD_T_W1]);
    const float* w2 = ref_ptr<const float>(c, op.t[S2K_SE_FC_BWD_T_W2]);
    float* dw1 = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DW1]);
    float* db1 = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DB1]);
    float* dw2 = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DW2]);
    float* db2 = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DB2]);
    float* dpool = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_DPOOL]);
    float* hs = ref_ptr<float>(c, op.t[S2K_SE_FC_BWD_T_HS]);
    CHECK_PTRS("se_fc_bwd", dgate, gate, hpre, pool, w1, w2, dw1, db1, dw2, db2, dpool, hs);
    const int B = op.d[S2K_SE_FC_BWD_D_B], C = op.d[S2K_SE_FC_BWD_D_C], Q = op.d[S2K_SE_FC_BWD_D_CSQ];
    const bool params = dw1 || db1 || dw2 || db2;        // none: SE_FC_WGRAD computes them (possibly on the side stream)
    if (!dgate || !gate || !hpre || !pool || !w1 || !w2 || !dpool || !hs || B <= 0 || (params && (!dw1 || !db1 || !dw2 || !db2))) {
        set_error("se_fc_bwd: bad args"); return S2K_EINVAL;
    }
    if (Q > 8192 || B > 65535) { set_error("se_fc_bwd: C/Q too large"); return S2K_EINVAL; }
    if ((size_t)(C + 1024) * 4 > 64000) { set_error("se_fc_bwd: C too large"); return S2K_EINVAL; }
    if (Q <= 64) {
        hipLaunchKernelGGL(se_fc_bwd_small_kernel, dim3(B), dim3(1024), (size_t)(C + 1024 + 64) * sizeof(float), c.stream, dgate, gate, hpre, w1, w2,
                           hs, dpool, C, Q);
    } else {
    hipLaunchKernelGGL(se_fc_bwd_a1_kernel, dim3(cdiv(Q, 64), B), dim3(1024), (C + 1024) * sizeof(float), c.stream, dgate, gate, hpre, w2, hs, B, C, Q);
    hipLaunchKernelGGL(se_fc_bwd_a2_kernel, dim3(cdiv(C, NTHREADS), B), dim3(NTHREADS), Q * sizeof(float), c.stream, dgate, gate, hpre, w1, dpool, B, C, Q);
    }
    if (!params) return S2K_OK;
    launch_se_fc_param_grads(dgate, hs, hpre, pool, dw1, db1, dw2, db2, B, C, Q, c.stream);
    return S2K_OK;
}

int launch_se_fc_wgrad(const S2kOp& op, const Ctx& c) {
    const float* dgp = ref_ptr<const float>(c, op.t[S2K_SE_FC_WGRAD_T_DGP]);
    const float* hs = ref_ptr<const float>(c, op.t[S2K_SE_FC_WGRAD_T_HS]);
    const float* dhp = ref_ptr<const float>(c, op.t[S2K_SE_FC_WGRAD_T_DHP]);
    const float* pool = ref_ptr<const float>(c, op.t[S2K_SE_FC_WGRAD_T_POOL]);
    float* dw1 = ref_ptr<float>(c, op.t[S2K_SE_FC_WGRAD_T_DW1]);
    float* db1 = ref_ptr<float>(c, op.t[S2K_SE_FC_WGRAD_T_DB1]);
    float* dw2 = ref_ptr<float>(c, op.t[S2K_SE_FC_WGRAD_T_DW2]);
    float* db2 = ref_ptr<float>(c, op.t[S2K_SE_FC_WGRAD_T_DB2]);
    CHECK_PTRS("se_fc_wgrad", dgp, hs, dhp, pool, dw1, db1, dw2, db2);
    const int B = op.d[S2K_SE_FC_WGRAD_D_B], C = op.d[S2K_SE_FC_WGRAD_D_C], Q = op.d[S2K_SE_FC_WGRAD_D_CSQ];
    if (!dgp || !hs || !dhp || !pool || !dw1 || !db1 || !dw2 || !db2 || B <= 0 || C <= 0 || Q <= 0) { set_error("se_fc_wgrad: bad args"); return S2K_EINVAL; }
    launch_se_fc_param_grads(dgp, hs, dhp, pool, dw1, db1, dw2, db2, B, C, Q, c.stream);
    return S2K_OK;
}

// ---------------- BN backward ------------------------------------------------------------------------------
// g' = (g * mulbc[b][c] * dcs[b] + addbc[b][c] * addscale) * act'(scale*y + shift); sums of g' and g'*xhat
template <bool VEC>
__global__ void __launch_bounds__(NTHREADS) bn_bwd_reduce_kernel(const float* g, const float* y, const float* bnv,
                                                                 const float* mulbc, const float* addbc, const float* noise,
                                                                 float* gout, double* stats2, int C, int HW, int64_t nplanes,
                                                                 int act, float keep, float addscale, int nrep) {
    Task t;
    if (!get_task(HW, nplanes, t)) return;
    const int lane = threadIdx.x & 63;
    const int c = (int)((uint32_t)t.plane % (uint32_t)C);
    const int b = (int)((uint32_t)t.plane / (uint32_t)C);
    const float scale = bnv[c], shift = bnv[C + c], mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
    float mul = mulbc ? mulbc[t.plane] : 1.0f;
    if (noise) mul *= floorf(keep + noise[b]) / keep;
    const float add = addbc ? addbc[t.plane] * addscale : 0.0f;
    const int64_t base = t.plane * HW + t.start;
    float s1 = 0.0f, s2 = 0.0f;
    if (VEC) {
        const int n4 = t.count >> 2;
        for (int i = lane; i < n4; i += 64) {
            float4 gv = reinterpret_cast<const float4*>(g + base)[i];
            const float4 yv = reinterpret_cast<const float4*>(y + base)[i];
            float* gp = &gv.x;
            const float* yp = &yv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = fmaf(gp[k], mul, add);
                if (act != S2K_PRO_NONE) v *= act_grad(fmaf(yp[k], scale, shift), act);
                gp[k] = v;
                s1 += v;
                s2 = fmaf(v, (yp[k] - mean) * invstd, s2);
            }
            reinterpret_cast<float4*>(gout + base)[i] = gv;
        }
    } else {
        for (int i = lane; i < t.count; i += 64) {
            const float yy = y[base + i];
            float v = fmaf(g[base + i], mul, add);
            if (act != S2K_PRO_NONE) v *= act_grad(fmaf(yy, scale, shift), act);
            gout[base + i] = v;
            s1 += v;
            s2 = fmaf(v, (yy - mean) * invstd, s2);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        double* st = stats2 + (int64_t)(blockIdx.x % nrep) * 2 * C;
        atomic_add_d(st + c, (double)s1);
        atomic_add_d(st + C + c, (double)s2);
    }
}

int launch_bn_bwd_reduce(const S2kOp& op, const Ctx& c) {
    const float* g = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_G]);
    const float* y = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_BNV]);
    const float* mulbc = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_MULBC]);
    const float* addbc = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_ADDBC]);
    const float* noise = ref_ptr<const float>(c, op.t[S2K_BN_BWD_REDUCE_T_NOISE]);
    float* gout = ref_ptr<float>(c, op.t[S2K_BN_BWD_REDUCE_T_GOUT]);
    double* st2 = ref_ptr<double>(c, op.t[S2K_BN_BWD_REDUCE_T_STATS2]);
    CHECK_PTRS("bn_bwd_reduce", g, y, bnv, mulbc, addbc, noise, gout, st2);
    const int B = op.d[S2K_BN_BWD_REDUCE_D_B], C = op.d[S2K_BN_BWD_REDUCE_D_C], HW = op.d[S2K_BN_BWD_REDUCE_D_HW];
    if (!g || !y || !bnv || !gout || !st2 || B <= 0 || C <= 0 || HW <= 0) { set_error("bn_bwd_reduce: bad args"); return S2K_EINVAL; }
    const int64_t nplanes = (int64_t)B * C;
    const unsigned blocks = task_blocks(HW, nplanes);
    const int act = op.d[S2K_BN_BWD_REDUCE_D_ACT];
    const float keep = op.f[S2K_BN_BWD_REDUCE_F_KEEP], addscale = op.f[S2K_BN_BWD_REDUCE_F_ADDSCALE];
    const int nrep = op.d[S2K_BN_BWD_REDUCE_D_NREP] > 0 ? op.d[S2K_BN_BWD_REDUCE_D_NREP] : 1;
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<true>), dim3(blocks), dim3(NTHREADS), 0, c.stream, g, y, bnv, mulbc, addbc, noise,
                           gout, st2, C, HW, nplanes, act, keep, addscale, nrep);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<false>), dim3(blocks), dim3(NTHREADS), 0, c.stream, g, y, bnv, mulbc, addbc, noise,
                           gout, st2, C, HW, nplanes, act, keep, addscale, nrep);
    return S2K_OK;
}

__global__ void bn_bwd_finalize_kernel(const double* st2, const float* gamma, const float* bnv, float* dgamma, float* dbeta,
                                       float* coef, int C, double count, int nrep) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = lane; r < nrep; r += 64) { s1 += st2[(int64_t)r * 2 * C + c]; s2 += st2[(int64_t)r * 2 * C + C + c]; }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    dgamma[c] += (float)s2;
    dbeta[c] += (float)s1;
    const double a = (double)gamma[c] * (double)bnv[3 * C + c];
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * s2 / count);
    coef[2 * C + c] = (float)(-a * s1 / count);
}

int launch_bn_bwd_finalize(const S2kOp& op, const Ctx& c) {
    const double* st2 = ref_ptr<const double>(c, op.t[S2K_BN_BWD_FINALIZE_T_STATS2]);
    const float* gamma = ref_ptr<const float>(c, op.t[S2K_BN_BWD_FINALIZE_T_GAMMA]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_BN_BWD_FINALIZE_T_BNV]);
    float* dgamma = ref_ptr<float>(c, op.t[S2K_BN_BWD_FINALIZE_T_DGAMMA]);
    float* dbeta = ref_ptr<float>(c, op.t[S2K_BN_BWD_FINALIZE_T_DBETA]);
    float* coef = ref_ptr<float>(c, op.t[S2K_BN_BWD_FINALIZE_T_COEF]);
    CHECK_PTRS("bn_bwd_finalize", st2, gamma, bnv, dgamma, dbeta, coef);
    const int C = op.d[S2K_BN_BWD_FINALIZE_D_C];
    if (!st2 || !gamma || !bnv || !dgamma || !dbeta || !coef || C <= 0) { set_error("bn_bwd_finalize: bad args"); return S2K_EINVAL; }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, c.stream, st2, gamma, bnv, dgamma, dbeta, coef, C,
                       (double)op.n[S2K_BN_BWD_FINALIZE_N_COUNT], op.d[S2K_BN_BWD_FINALIZE_D_NREP] > 0 ? op.d[S2K_BN_BWD_FINALIZE_D_NREP] : 1);
    return S2K_OK;
}

// the per-channel work of BN_BWD_FINALIZE done by the consumer: sums of g' and g'*xhat over the replicas -> coefficients;
// the wave that owns the first chunk of sample 0 publishes the parameter gradients
struct BnFuse {
    const double* st2;
    const float* gamma;
    float *dgamma, *dbeta;
    double inv_count;
    int nrep;
    const float *mulbc, *addbc;   // MODE 3: g' = (GP * mul + add * addscale) * silu'(u) is recomputed here
    float addscale;
    const float* ps;              // MODE 3: SE_BN_SUMS's plane sums [4][B][C]; the sums over the batch are formed here (SE_BN_COMBINE)
    int B;
};

// the channel's two BN-backward sums: from the STATS2 replicas, or (fz.ps) combined from the SE backward's plane sums - the same
// lane order in every wave, so all waves of a channel agree bit for bit
__device__ __forceinline__ void bn_fuse_sums(const BnFuse& fz, int C, int c, int lane, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    if (fz.ps) {
        const int64_t np = (int64_t)fz.B * C;
        for (int b = lane; b < fz.B; b += 64) {
            const int64_t pl = (int64_t)b * C + c;
            const double mul = fz.mulbc ? (double)fz.mulbc[pl] : 1.0, add = fz.addbc ? (double)fz.addbc[pl] * (double)fz.addscale : 0.0;
            s1 += mul * (double)fz.ps[pl] + add * (double)fz.ps[np + pl];
            s2 += mul * (double)fz.ps[2 * np + pl] + add * (double)fz.ps[3 * np + pl];
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
    } else if (fz.nrep <= 8) {   // wave-uniform addresses: scalar loads
        for (int r = 0; r < fz.nrep; ++r) { s1 += fz.st2[(int64_t)r * 2 * C + c]; s2 += fz.st2[(int64_t)r * 2 * C + C + c]; }
    } else {
        for (int r = lane; r < fz.nrep; r += 64) { s1 += fz.st2[(int64_t)r * 2 * C + c]; s2 += fz.st2[(int64_t)r * 2 * C + C + c]; }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
    }
}

// MODE 0: dy = A*gp + Bq*xhat + Cq (BN_BWD_APPLY; MODE 2 = the same with the coefficients computed here from STATS2;
// MODE 3 = MODE 2 with gp = (a*mul + add) * silu'(scale*y + shift) recomputed per element, see SE_BN_SUMS);
// MODE 1: xout = (scale*y+shift)*dcs[b] + ident (BN_RESIDUAL)
// OUT16 (MODE 2, VEC): the result is stored as bf16 (round to nearest even, `v_cvt_pk_bf16_f32`) - dY of a bf16-mixed plan whose only
// readers are bf16 MFMA stages that would round exactly these values themselves (opdefs CONV.X1_BF16)
__device__ __forceinline__ uint32_t pm_pk_bf16(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

template <int MODE, bool VEC, bool OUT16 = false>
__global__ void __launch_bounds__(NTHREADS) plane_map_kernel(const float* a, const float* y, const float* bnv, const float* coef,
                                                             const float* noise, float* out, int C, int HW, int64_t nplanes,
                                                             float keep, const BnFuse fz, const BnFold fold = BnFold{}) {
    Task t;
    if (!get_task(HW, nplanes, t)) return;
    const int lane = threadIdx.x & 63;
    const int c = (int)((uint32_t)t.plane % (uint32_t)C);
    const int64_t base = t.plane * HW + t.start;
    // the first element group is requested BEFORE the per-channel coefficients are formed (replica / plane sums in f64: a few
    // dependent L2 round trips), so the stream is already in flight when the arithmetic needs it
    float4 av0 = make_float4(0, 0, 0, 0), yv0 = make_float4(0, 0, 0, 0);
    if (VEC && lane < (t.count >> 2)) {
        if (a) av0 = reinterpret_cast<const float4*>(a + base)[lane];
        yv0 = reinterpret_cast<const float4*>(y + base)[lane];
    }
    float k0, k1, k2;  // out = k0*a + k1*y + k2
    float gmul = 1.0f, gadd = 0.0f, bscale = 1.0f, bshift = 0.0f;
    if (MODE == 3) {
        gmul = fz.mulbc ? fz.mulbc[t.plane] : 1.0f;
        gadd = fz.addbc ? fz.addbc[t.plane] * fz.addscale : 0.0f;
        bscale = bnv[c];
        bshift = bnv[C + c];
    }
    if (MODE >= 2) {
        double s1, s2;
        bn_fuse_sums(fz, C, c, lane, s1, s2);
        const float mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
        const double aa = (double)fz.gamma[c] * (double)invstd;
        const float A = (float)aa, Bq = (float)(-aa * s2 * fz.inv_count), Cq = (float)(-aa * s1 * fz.inv_count);   // no f64 divide per wave
        k0 = A; k1 = Bq * invstd; k2 = Cq - Bq * invstd * mean;
        if (t.plane < C && t.start == 0 && lane == 0) {
            fz.dgamma[c] += (float)s2;
            fz.dbeta[c] += (float)s1;
        }
    } else if (MODE == 0) {
        const float mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
        const float A = coef[c], Bq = coef[C + c], Cq = coef[2 * C + c];
        k0 = A; k1 = Bq * invstd; k2 = Cq - Bq * invstd * mean;
    } else {
        float dcs = 1.0f;
        if (noise) dcs = floorf(keep + noise[(int)(t.plane / C)]) / keep;
        float sc, sh;
        if (fold.stats) bn_fold_wave(fold, C, c, t.plane < C && t.start == 0, sc, sh);   // BN_FINALIZE folded in (one wave per task)
        else { sc = bnv[c]; sh = bnv[C + c]; }
        k0 = a ? 1.0f : 0.0f; k1 = sc * dcs; k2 = sh * dcs;
    }
    if (VEC) {
        const int n4 = t.count >> 2;
        for (int i = lane; i < n4; i += 64) {
            float4 av = av0, yv = yv0;
            if (i + 64 < n4) {                      // next group in flight while this one is computed and stored
                if (a) av0 = reinterpret_cast<const float4*>(a + base)[i + 64];
                yv0 = reinterpret_cast<const float4*>(y + base)[i + 64];
            }
            if (MODE == 3) {
                av.x = fmaf(av.x, gmul, gadd) * act_grad(fmaf(yv.x, bscale, bshift), S2K_PRO_SILU);
                av.y = fmaf(av.y, gmul, gadd) * act_grad(fmaf(yv.y, bscale, bshift), S2K_PRO_SILU);
                av.z = fmaf(av.z, gmul, gadd) * act_grad(fmaf(yv.z, bscale, bshift), S2K_PRO_SILU);
                av.w = fmaf(av.w, gmul, gadd) * act_grad(fmaf(yv.w, bscale, bshift), S2K_PRO_SILU);
            }
            float4 o;
            o.x = fmaf(k0, av.x, fmaf(k1, yv.x, k2)); o.y = fmaf(k0, av.y, fmaf(k1, yv.y, k2));
            o.z = fmaf(k0, av.z, fmaf(k1, yv.z, k2)); o.w = fmaf(k0, av.w, fmaf(k1, yv.w, k2));
            if constexpr (OUT16) reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + base)[i] = make_uint2(pm_pk_bf16(o.x, o.y), pm_pk_bf16(o.z, o.w));
            else reinterpret_cast<float4*>(out + base)[i] = o;
        }
    } else {
        for (int i = lane; i < t.count; i += 64) {
            float av = a ? a[base + i] : 0.0f;
            if (MODE == 3) av = fmaf(av, gmul, gadd) * act_grad(fmaf(y[base + i], bscale, bshift), S2K_PRO_SILU);
            out[base + i] = fmaf(k0, av, fmaf(k1, y[base + i], k2));
        }
    }
}

// BN_BWD_APPLY (fused finalize, MODE 2 / 3 of plane_map_kernel) on small planes (<= 64 elements: the 8x8 stages, 1824 - 3072
// channels): one wave per CHANNEL (and batch chunk) forms the three coefficients once - the replica sums, two f64 products -
// and walks the batch with 16 lanes per plane, four planes per pass, the loads of four passes in flight.  With one wave per
// 64-element plane the coefficient arithmetic was most of the kernel (23 us for a 15 MB tensor).
template <int MODE, bool OUT16 = false>
__global__ void __launch_bounds__(NTHREADS) bn_bwd_apply_small_kernel(const float* a, const float* y, const float* bnv, float* out,
                                                                      int B, int C, int HW, int lpp, const BnFuse fz) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1, s2;
    bn_fuse_sums(fz, C, c, lane, s1, s2);
    const float mean = bnv[2 * C + c], invstd = bnv[3 * C + c];
    const double aa = (double)fz.gamma[c] * (double)invstd;
    const float A = (float)aa, Bq = (float)(-aa * s2 * fz.inv_count), Cq = (float)(-aa * s1 * fz.inv_count);
    const float k0 = A, k1 = Bq * invstd, k2 = Cq - Bq * invstd * mean;
    if (blockIdx.y == 0 && lane == 0) {
        fz.dgamma[c] += (float)s2;
        fz.dbeta[c] += (float)s1;
    }
    const float bscale = bnv[c], bshift = bnv[C + c];
    const int pp = 64 / lpp, sub = lane / lpp, li = lane & (lpp - 1);
    if (4 * li >= HW) return;
    const int bper = (B + gridDim.y - 1) / gridDim.y, b_lo = blockIdx.y * bper, b_hi = min(B, b_lo + bper);
    constexpr int U = 4;
    for (int b0 = b_lo + sub; b0 < b_hi; b0 += pp * U) {
        float4 av[U], yv[U];
        float gmul[U], gadd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = min(b0 + u * pp, b_hi - 1);                   // clamped: loads never sit under a condition
            const int64_t plane = (int64_t)b * C + c, off = plane * HW + 4 * li;
            av[u] = *reinterpret_cast<const float4*>(a + off);
            yv[u] = *reinterpret_cast<const float4*>(y + off);
            gmul[u] = 1.0f;
            gadd[u] = 0.0f;
            if (MODE == 3) {
                if (fz.mulbc) gmul[u] = fz.mulbc[plane];
                if (fz.addbc) gadd[u] = fz.addbc[plane] * fz.addscale;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * pp;
            if (b >= b_hi) break;
            float4 g = av[u];
            const float4 yy = yv[u];
            if (MODE == 3) {
                g.x = fmaf(g.x, gmul[u], gadd[u]) * act_grad(fmaf(yy.x, bscale, bshift), S2K_PRO_SILU);
                g.y = fmaf(g.y, gmul[u], gadd[u]) * act_grad(fmaf(yy.y, bscale, bshift), S2K_PRO_SILU);
                g.z = fmaf(g.z, gmul[u], gadd[u]) * act_grad(fmaf(yy.z, bscale, bshift), S2K_PRO_SILU);
                g.w = fmaf(g.w, gmul[u], gadd[u]) * act_grad(fmaf(yy.w, bscale, bshift), S2K_PRO_SILU);
            }
            float4 o;
            o.x = fmaf(k0, g.x, fmaf(k1, yy.x, k2)); o.y = fmaf(k0, g.y, fmaf(k1, yy.y, k2));
            o.z = fmaf(k0, g.z, fmaf(k1, yy.z, k2)); o.w = fmaf(k0, g.w, fmaf(k1, yy.w, k2));
            if constexpr (OUT16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + ((int64_t)b * C + c) * HW + 4 * li) = make_uint2(pm_pk_bf16(o.x, o.y), pm_pk_bf16(o.z, o.w));
            else *reinterpret_cast<float4*>(out + ((int64_t)b * C + c) * HW + 4 * li) = o;
        }
    }
}

template <int MODE, bool OUT16 = false>
static bool launch_bn_bwd_apply_small(const float* a, const float* y, const float* bnv, float* out, int B, int C, int HW,
                                      hipStream_t st, const BnFuse& fz) {
    if (HW > 64 || (HW & 3) || B < 2) return false;
    int lpp = 1;
    while (4 * lpp < HW) lpp <<= 1;
    const int bsplit = std::max(1, std::min(cdiv(B, 64 / lpp), cdiv(2048, C)));
    hipLaunchKernelGGL((bn_bwd_apply_small_kernel<MODE, OUT16>), dim3(cdiv(C, 4), bsplit), dim3(NTHREADS), 0, st, a, y, bnv, out, B, C, HW, lpp, fz);
    return true;
}

// BN_RESIDUAL on small planes: one wave per channel walks the batch (see se_pool_small_kernel)
__global__ void __launch_bounds__(NTHREADS) bn_residual_small_kernel(const float* ident, const float* y, const float* bnv, const float* noise,
                                                                     float* out, int B, int C, int HW, float keep, int lpp,
                                                                     const BnFold fold) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float sc, sh;
    if (fold.stats) bn_fold_wave(fold, C, c, blockIdx.y == 0, sc, sh);
    else { sc = bnv[c]; sh = bnv[C + c]; }
    const int pp = 64 / lpp, sub = lane / lpp, li = lane & (lpp - 1);
    if (4 * li >= HW) return;
    const int bper = (B + gridDim.y - 1) / gridDim.y, b_lo = blockIdx.y * bper;
    B = min(B, b_lo + bper);
    for (int b = b_lo + sub; b < B; b += pp) {
        float dcs = 1.0f;
        if (noise) dcs = floorf(keep + noise[b]) / keep;
        const float k1 = sc * dcs, k2 = sh * dcs;
        const int64_t off = ((int64_t)b * C + c) * HW + 4 * li;
        const float4 yv = *reinterpret_cast<const float4*>(y + off);
        float4 av = make_float4(0, 0, 0, 0);
        if (ident) av = *reinterpret_cast<const float4*>(ident + off);
        float4 o;
        o.x = av.x + fmaf(k1, yv.x, k2); o.y = av.y + fmaf(k1, yv.y, k2);
        o.z = av.z + fmaf(k1, yv.z, k2); o.w = av.w + fmaf(k1, yv.w, k2);
        *reinterpret_cast<float4*>(out + off) = o;
    }
}

template <int MODE>
static void launch_plane_map(const float* a, const float* y, const float* bnv, const float* coef, const float* noise, float* out,
                             int B, int C, int HW, float keep, hipStream_t st, const BnFuse& fz = BnFuse{}, const BnFold& fold = BnFold{}) {
    const int64_t nplanes = (int64_t)B * C;
    const unsigned blocks = task_blocks(HW, nplanes);
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((plane_map_kernel<MODE, true>), dim3(blocks), dim3(NTHREADS), 0, st, a, y, bnv, coef, noise, out, C, HW, nplanes, keep, fz, fold);
    else
        hipLaunchKernelGGL((plane_map_kernel<MODE, false>), dim3(blocks), dim3(NTHREADS), 0, st, a, y, bnv, coef, noise, out, C, HW, nplanes, keep, fz, fold);
}

int launch_bn_bwd_apply(const S2kOp& op, const Ctx& c) {
    const float* gp = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_GP]);
    const float* y = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_BNV]);
    const float* coef = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_COEF]);
    float* dy = ref_ptr<float>(c, op.t[S2K_BN_BWD_APPLY_T_DY]);
    BnFuse fz{};
    fz.st2 = ref_ptr<const double>(c, op.t[S2K_BN_BWD_APPLY_T_STATS2]);
    fz.gamma = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_GAMMA]);
    fz.dgamma = ref_ptr<float>(c, op.t[S2K_BN_BWD_APPLY_T_DGAMMA]);
    fz.dbeta = ref_ptr<float>(c, op.t[S2K_BN_BWD_APPLY_T_DBETA]);
    const double count = (double)op.n[S2K_BN_BWD_APPLY_N_COUNT];
    fz.inv_count = (count > 0 && !op.d[S2K_BN_BWD_APPLY_D_EVAL]) ? 1.0 / count : 0.0;   // EVAL: frozen statistics, Bq = Cq = 0
    fz.nrep = op.d[S2K_BN_BWD_APPLY_D_NREP] > 0 ? op.d[S2K_BN_BWD_APPLY_D_NREP] : 1;
    CHECK_PTRS("bn_bwd_apply", gp, y, bnv, coef, dy, fz.st2, fz.gamma, fz.dgamma, fz.dbeta);
    if (!gp || !y || !bnv || !dy) { set_error("bn_bwd_apply: bad args"); return S2K_EINVAL; }
    const int B = op.d[S2K_BN_BWD_APPLY_D_B], C = op.d[S2K_BN_BWD_APPLY_D_C], HW = op.d[S2K_BN_BWD_APPLY_D_HW];
    if (coef) {
        if (op.d[S2K_BN_BWD_APPLY_D_OUT_BF16]) { set_error("bn_bwd_apply: OUT_BF16 needs the fused form (no COEF)"); return S2K_EINVAL; }
        launch_plane_map<0>(gp, y, bnv, coef, nullptr, dy, B, C, HW, 1.0f, c.stream);
        return S2K_OK;
    }
    // no COEF table: the BN_BWD_FINALIZE arithmetic is done here (one launch less per BatchNorm)
    fz.ps = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_PS]);
    fz.B = B;
    if ((!fz.st2 && !fz.ps) || !fz.gamma || !fz.dgamma || !fz.dbeta || count <= 0) { set_error("bn_bwd_apply: fused form needs STATS2 (or PS), GAMMA, DGAMMA, DBETA, COUNT"); return S2K_EINVAL; }
    fz.mulbc = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_MULBC]);
    fz.addbc = ref_ptr<const float>(c, op.t[S2K_BN_BWD_APPLY_T_ADDBC]);
    fz.addscale = op.f[S2K_BN_BWD_APPLY_F_ADDSCALE];
    CHECK_PTRS("bn_bwd_apply", fz.mulbc, fz.addbc, fz.ps);
    if (fz.ps && op.d[S2K_BN_BWD_APPLY_D_ACT] != S2K_PRO_SILU) { set_error("bn_bwd_apply: PS belongs to the recomputing (SiLU) form"); return S2K_EINVAL; }
    const bool out16 = op.d[S2K_BN_BWD_APPLY_D_OUT_BF16] != 0;
    if (out16 && (op.d[S2K_BN_BWD_APPLY_D_ACT] != S2K_PRO_NONE || fz.mulbc || fz.addbc || (HW & 3) || dy == gp)) {
        set_error("bn_bwd_apply: OUT_BF16 is for the plain fused form, HW % 4 == 0, out of place"); return S2K_EINVAL;
    }
    if (out16) {
        if (launch_bn_bwd_apply_small<2, true>(gp, y, bnv, dy, B, C, HW, c.stream, fz)) return S2K_OK;
        const int64_t nplanes = (int64_t)B * C;
        hipLaunchKernelGGL((plane_map_kernel<2, true, true>), dim3(task_blocks(HW, nplanes)), dim3(NTHREADS), 0, c.stream, gp, y, bnv,
                           (const float*)nullptr, (const float*)nullptr, dy, C, HW, nplanes, 1.0f, fz, BnFold{});
        return S2K_OK;
    }
    if (op.d[S2K_BN_BWD_APPLY_D_ACT] == S2K_PRO_NONE && !fz.mulbc && !fz.addbc) {
        if (launch_bn_bwd_apply_small<2>(gp, y, bnv, dy, B, C, HW, c.stream, fz)) return S2K_OK;
        launch_plane_map<2>(gp, y, bnv, nullptr, nullptr, dy, B, C, HW, 1.0f, c.stream, fz);
        return S2K_OK;
    }
    // GP holds the raw upstream gradient: g' = (GP * MULBC + ADDBC * ADDSCALE) * silu'(u) is recomputed per element
    if (op.d[S2K_BN_BWD_APPLY_D_ACT] != S2K_PRO_SILU) { set_error("bn_bwd_apply: the recomputing form is SiLU only"); return S2K_EINVAL; }
    if (launch_bn_bwd_apply_small<3>(gp, y, bnv, dy, B, C, HW, c.stream, fz)) return S2K_OK;
    launch_plane_map<3>(gp, y, bnv, nullptr, nullptr, dy, B, C, HW, 1.0f, c.stream, fz);
    return S2K_OK;
}

int launch_bn_residual(const S2kOp& op, const Ctx& c) {
    const float* y = ref_ptr<const float>(c, op.t[S2K_BN_RESIDUAL_T_Y]);
    const float* bnv = ref_ptr<const float>(c, op.t[S2K_BN_RESIDUAL_T_BNV]);
    const float* ident = ref_ptr<const float>(c, op.t[S2K_BN_RESIDUAL_T_IDENT]);
    const float* noise = ref_ptr<const float>(c, op.t[S2K_BN_RESIDUAL_T_NOISE]);
    float* xout = ref_ptr<float>(c, op.t[S2K_BN_RESIDUAL_T_XOUT]);
    CHECK_PTRS("bn_residual", y, bnv, ident, noise, xout);
    if (!y || !bnv || !xout) { set_error("bn_residual: bad args"); return S2K_EINVAL; }
    BnFold fold;
    if (int e = fill_bn_fold(fold, c, &op.t[S2K_BN_RESIDUAL_T_FSTATS], op.n[S2K_BN_RESIDUAL_N_FCOUNT], op.d[S2K_BN_RESIDUAL_D_FNREP],
                             op.f[S2K_BN_RESIDUAL_F_FEPS], op.f[S2K_BN_RESIDUAL_F_FMOM], const_cast<float*>(bnv), "bn_residual")) return e;
    {
        const int B = op.d[S2K_BN_RESIDUAL_D_B], C = op.d[S2K_BN_RESIDUAL_D_C], HW = op.d[S2K_BN_RESIDUAL_D_HW];
        if (HW <= 256 && (HW & 3) == 0 && B > 1) {
            int lpp = 1;
            while (4 * lpp < HW) lpp <<= 1;
            const int bsplit = std::max(1, std::min(cdiv(B, 64 / lpp), cdiv(2048, C)));
            hipLaunchKernelGGL(bn_residual_small_kernel, dim3(cdiv(C, 4), bsplit), dim3(NTHREADS), 0, c.stream, ident, y, bnv, noise, xout, B, C, HW,
                               op.f[S2K_BN_RESIDUAL_F_KEEP], lpp, fold);
            return S2K_OK;
        }
    }
    launch_plane_map<1>(ident, y, bnv, nullptr, noise, xout, op.d[S2K_BN_RESIDUAL_D_B], op.d[S2K_BN_RESIDUAL_D_C],
                        op.d[S2K_BN_RESIDUAL_D_HW], op.f[S2K_BN_RESIDUAL_F_KEEP], c.stream, BnFuse{}, fold);
    return S2K_OK;
}

// ---------------- space-to-depth (pixel unshuffle) of a ConvTranspose output gradient --------------------------------
// one thread: 8 consecutive high-resolution pixels of one row (two float4 loads) -> float4 to channel (c, dy, 0) and
// float4 to (c, dy, 1); lanes run along the row, reads and writes are full 16-byte vectors
__global__ void __launch_bounds__(NTHREADS) space_to_depth_kernel(const float* x, float* y, int C, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;   // over (b, c, dy, yy, xq), xq = groups of 4 low-res columns
    if (i >= total) return;
    const int wq = W >> 2;
    const int xq = (int)(i % wq);
    int64_t r = i / wq;
    const int yy = (int)(r % H); r /= H;
    const int dy = (int)(r & 1); r >>= 1;                     // r = b * C + c
    const float* src = x + (r * 2 * H + 2 * yy + dy) * (int64_t)(2 * W) + 8 * xq;
    const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
    float* dst = y + ((r * 4 + dy * 2) * H + yy) * (int64_t)W + 4 * xq;
    *reinterpret_cast<float4*>(dst) = make_float4(a.x, a.z, b.x, b.z);
    *reinterpret_cast<float4*>(dst + (int64_t)H * W) = make_float4(a.y, a.w, b.y, b.w);
}

__global__ void space_to_depth_scalar_kernel(const float* x, float* y, int C, int H, int W, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {   // over Y elements
        const int xx = (int)(i % W);
        int64_t r = i / W;
        const int yy = (int)(r % H); r /= H;
        const int k = (int)(r & 3); r >>= 2;
        y[i] = x[(r * 2 * H + 2 * yy + (k >> 1)) * (int64_t)(2 * W) + 2 * xx + (k & 1)];
    }
}

int launch_space_to_depth(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_SPACE_TO_DEPTH_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_SPACE_TO_DEPTH_T_Y]);
    CHECK_PTRS("space_to_depth", x, y);
    const int B = op.d[S2K_SPACE_TO_DEPTH_D_B], C = op.d[S2K_SPACE_TO_DEPTH_D_C], H = op.d[S2K_SPACE_TO_DEPTH_D_H], W = op.d[S2K_SPACE_TO_DEPTH_D_W];
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("space_to_depth: bad args"); return S2K_EINVAL; }
    if ((W & 3) == 0) {
        const int64_t total = (int64_t)B * C * 2 * H * (W >> 2);
        hipLaunchKernelGGL(space_to_depth_kernel, dim3((unsigned)cdiv64(total, NTHREADS)), dim3(NTHREADS), 0, c.stream, x, y, C, H, W, total);
    } else {
        const int64_t total = (int64_t)B * C * 4 * H * W;
        hipLaunchKernelGGL(space_to_depth_scalar_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65536)), dim3(256), 0, c.stream, x, y, C, H, W, total);
    }
    return S2K_OK;
}

// ---------------- patch columns of a strided dense conv (IM2COL) ----------------------------------------------------------------
// one thread: 4 consecutive output columns of one (b, k, yo) row -> one float4 store; the strided reads of neighbouring lanes fall
// into the same cache lines.  Bounds-checked buffer loads give the zero padding.
__global__ void __launch_bounds__(NTHREADS) im2col_kernel(const float* x, float* y, int C, int H, int W, int KH, int KW, int S, int PT, int PL,
                                                          int HO, int WO, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;     // over (b, k, yo, xq)
    if (i >= total) return;
    const int wq = (WO + 3) >> 2;
    const int xq = (int)(i % wq);
    int64_t r = i / wq;
    const int yo = (int)(r % HO); r /= HO;
    const int K = C * KH * KW;
    const int k = (int)(r % K);
    const int64_t b = r / K;
    const int kx = k % KW, ky = (k / KW) % KH, c = k / (KW * KH);
    const int iy = yo * S + ky - PT;
    const float* src = x + ((b * C + c) * H) * (int64_t)W;             // the (b, c) plane: offsets below stay inside 32 bits (H * W * 4 < 2 GiB)
    const rsrc_t rs = make_rsrc(src, (int64_t)H * W * 4);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ix = (4 * xq + j) * S + kx - PL;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        v[j] = bload(rs, ok ? (uint32_t)(iy * W + ix) * 4u : BUF_OOB);
    }
    float* dst = y + ((b * K + k) * HO + yo) * (int64_t)WO + 4 * xq;
    if ((WO & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * xq + j < WO) dst[j] = v[j];
}

int launch_im2col(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_IM2COL_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_IM2COL_T_Y]);
    CHECK_PTRS("im2col", x, y);
    const int32_t* d = op.d;
    const int B = d[S2K_IM2COL_D_B], C = d[S2K_IM2COL_D_C], H = d[S2K_IM2COL_D_H], W = d[S2K_IM2COL_D_W], KH = d[S2K_IM2COL_D_KH], KW = d[S2K_IM2COL_D_KW];
    const int S = d[S2K_IM2COL_D_STRIDE], PT = d[S2K_IM2COL_D_PAD_T], PL = d[S2K_IM2COL_D_PAD_L], HO = d[S2K_IM2COL_D_HO], WO = d[S2K_IM2COL_D_WO];
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || S <= 0 || HO <= 0 || WO <= 0 || PT < 0 || PL < 0 ||
        (int64_t)H * W * 4 >= 0x7ffffff0ll) { set_error("im2col: bad args"); return S2K_EINVAL; }
    const int64_t total = (int64_t)B * C * KH * KW * HO * ((WO + 3) >> 2);
    if (cdiv64(total, NTHREADS) > 0x7fffffffll) { set_error("im2col: grid too large"); return S2K_EINVAL; }
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)cdiv64(total, NTHREADS)), dim3(NTHREADS), 0, c.stream, x, y, C, H, W, KH, KW, S, PT, PL, HO, WO, total);
    return S2K_OK;
}

// ---------------- zero-insertion upsampling (data gradient of a strided dense conv; only planned for input gradients) ------
__global__ void upsample_zero_kernel(const float* x, float* y, int H, int W, int S, int HO, int WO, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {   // over Y elements
        const int xx = (int)(i % WO);
        int64_t r = i / WO;
        const int yy = (int)(r % HO);
        r /= HO;                                                  // r = b * C + c
        const int ys = yy / S, xs = xx / S;
        const bool on = (yy - ys * S) == 0 && (xx - xs * S) == 0 && ys < H && xs < W;
        y[i] = on ? x[(r * H + ys) * (int64_t)W + xs] : 0.0f;
    }
}

int launch_upsample_zero(const S2kOp& op, const Ctx& c) {
    const float* x = ref_ptr<const float>(c, op.t[S2K_UPSAMPLE_ZERO_T_X]);
    float* y = ref_ptr<float>(c, op.t[S2K_UPSAMPLE_ZERO_T_Y]);
    CHECK_PTRS("upsample_zero", x, y);
    const int B = op.d[S2K_UPSAMPLE_ZERO_D_B], C = op.d[S2K_UPSAMPLE_ZERO_D_C], H = op.d[S2K_UPSAMPLE_ZERO_D_H], W = op.d[S2K_UPSAMPLE_ZERO_D_W];
    const int S = op.d[S2K_UPSAMPLE_ZERO_D_S], HO = op.d[S2K_UPSAMPLE_ZERO_D_HO], WO = op.d[S2K_UPSAMPLE_ZERO_D_WO];
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || S <= 0 || HO <= 0 || WO <= 0) { set_error("upsample_zero: bad args"); return S2K_EINVAL; }
    const int64_t total = (int64_t)B * C * HO * WO;
    hipLaunchKernelGGL(upsample_zero_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65536)), dim3(256), 0, c.stream, x, y, H, W, S, HO, WO, total);
    return S2K_OK;
}

// ---------------- fused Adam (L2-coupled weight decay; torch.optim.Adam semantics) -------------------------------
// Same operation order as torch's single-tensor Adam (the optimiser the reference configures,
// /root/reference/src/train_segmentation.py:109-115):  g' = g + wd*p;  m = lerp(m, g', 1-b1);  v = b2*v + (1-b2)*g'*g';
// denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr/(1-b1^t)) * m/denom.  The bias corrections are computed in double on the
// host (Python does `1 - beta ** step` in double).  One float4 per lane: the pass streams 4 reads + 3 writes per parameter.
struct AdamK {
    float lr_c1;      // lr / (1 - b1^t)
    float sq_c2;      // sqrt(1 - b2^t)
    float b1w;        // 1 - b1
    float b2, b2w;    // b2, 1 - b2
    float eps, wd;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamK k) {
    if (k.wd != 0.0f) g = fmaf(k.wd, p, g);
    m = fmaf(k.b1w, g - m, m);
    v = fmaf(k.b2w * g, g, k.b2 * v);
    const float denom = sqrtf(v) / k.sq_c2 + k.eps;
    p = fmaf(-k.lr_c1, m / denom, p);
}
__global__ void __launch_bounds__(256) adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, const AdamK k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        adam_one(pv.x, gv.x, mv.x, vv.x, k); adam_one(pv.y, gv.y, mv.y, vv.y, k);
        adam_one(pv.z, gv.z, mv.z, vv.z, k); adam_one(pv.w, gv.w, mv.w, vv.w, k);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], k);
}
__global__ void adam_scalar_kernel(float* p, const float* g, float* m, float* v, int64_t n, const AdamK k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], k);
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps, double wd,
                int step, hipStream_t st) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) { set_error("adam: bad args"); return S2K_EINVAL; }
    const double c1 = 1.0 - pow(b1, (double)step), c2 = 1.0 - pow(b2, (double)step);
    AdamK k;
    k.lr_c1 = (float)(lr / c1);
    k.sq_c2 = (float)sqrt(c2);
    k.b1w = (float)(1.0 - b1);
    k.b2 = (float)b2;
    k.b2w = (float)(1.0 - b2);
    k.eps = (float)eps;
    k.wd = (float)wd;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (aligned) {
        const int blocks = (int)std::min<int64_t>(cdiv64(cdiv64(n, 4), 256), 8192);
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, k);
    } else {
        const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 8192);
        hipLaunchKernelGGL(adam_scalar_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, k);
    }
    return S2K_OK;
}

}  // namespace s2k
