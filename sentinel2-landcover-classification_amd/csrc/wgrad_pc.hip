// Weight gradients of the dense convs, producer / consumer form (the MFMA-bound shapes: 3x3 stride-1 convs on 64-pixel
// tiles, prologue-light 1x1 convs / Linears on 64..128-channel tiles).  Same arithmetic as wgrad.hip
// (dW[tap][m][c] += sum_pix P[m][pix] * Qpro[c][pix + tap], MFMA 32x32x2 with k = two consecutive pixels), replaces the
// same reference code (ATen convolution_backward grad_weight for efficientnet_unet.py:168-176,319-372 and every Linear of
// timm's Block, prithvi.py:162-183); what changes is WHO does what:
//
//   * 8 waves per workgroup.  Waves 0-3 (one per SIMD) are CONSUMERS: their instruction stream is LDS reads + MFMAs only,
//     fully unrolled over the pixel pairs of a tile with every LDS offset an instruction immediate (tile geometry is a
//     template parameter).  Measured alone they run at 93-97 % of the MFMA issue rate.
//   * Waves 4-7 (the second wave of each SIMD) are PRODUCERS: global -> registers -> (BatchNorm / activation prologue) ->
//     LDS for the NEXT tile into the other half of a double-buffered LDS image, then they park at the barrier.
//     Beside a wave that issues f32 MFMAs back to back, the partner wave gets roughly ONE instruction issued per MFMA
//     (measured with in-kernel stamps: ~60 cycles per producer instruction, whatever the priorities), so what bounds a
//     producer is its INSTRUCTION COUNT per tile, not bytes: the LDS image is therefore pixel-major ([pixel][channel], rows
//     padded to channel count + 4 floats), a producer thread owns channel QUADS (its four channel values of one pixel go out
//     in one ds_write_b128; scale / shift of its channels live in registers), and ReLU + zero padding is one v_med3_f32.
//     With channels on the consumer's lanes every operand read is 32 consecutive words per half-wave (conflict-free).
//   * One barrier per tile.  Producer: P(0) | P(1) | P(2) ...; consumer: | C(0) | C(1) ... (| = barrier): C(k) reads buffer
//     k & 1 after the barrier that follows P(k); P(k+2) overwrites it after the barrier that follows C(k).
//   * Workgroup ids are remapped so that the workgroups of one pixel split (which read the same P / Q rows) share an XCD's
//     L2 (L2 hit rate 22 % -> 88 %, fabric reads halved: profiles/r02_*).
#include <algorithm>

#include "common.h"
#include "wgrad.h"

namespace s2k {

// In-kernel stamps: only in the stamps build (-DS2K_TUNING -DS2K_DMA_STAMPS, libs2k_stamps.so).  They used to be part of every tuning
// build and made wgrad_pc_kernel 5 - 8 % slower there than in the shipped library, which skewed every A/B run against it.
#if defined(S2K_TUNING) && defined(S2K_DMA_STAMPS)
__device__ unsigned long long g_wg_dbg[8];      // {consumer barrier-wait cycles, compute cycles, tiles, producer wait, producer work}
#define WG_STAMP() __builtin_amdgcn_s_memtime()
#define WG_DBG_ADD(i, v) do { dbg_acc[i] += (unsigned long long)(v); } while (0)
#define WG_DBG_DECL() unsigned long long dbg_acc[5] = {0, 0, 0, 0, 0}
#define WG_DBG_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 5; ++i_) if (dbg_acc[i_]) atomicAdd(&g_wg_dbg[i_], dbg_acc[i_]); } while (0)
#else
#define WG_STAMP() 0ull
#define WG_DBG_ADD(i, v) do { } while (0)
#define WG_DBG_DECL() do { } while (0)
#define WG_DBG_FLUSH() do { } while (0)
#endif

// ReLU / identity prologue + validity in as few vector instructions as possible (`bound` = +inf for a valid element, 0 for
// padding; values of padding elements were loaded as 0)
template <int PRO>
__device__ __forceinline__ float pro_masked(float v, float sc, float sh, float bound, float keep) {
    if (PRO == S2K_PRO_NONE) return v;
    if (PRO == S2K_PRO_RELU) return __builtin_amdgcn_fmed3f(fmaf(v, sc, sh), 0.0f, bound);   // clamp to [0, bound]
    return apply_pro_c<PRO>(v, sc, sh) * keep;
}

// Consumer waves are arranged WVM x WVN x WVK: WVK > 1 splits the pixel pairs of a tile over waves (thin layers: a 32-channel side
// leaves only one 32 x 32 tile per tap, so the four waves share it and add their partial sums at the end).
template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK, int NPJ, int R, int XWE, int PROP, int PROQ>
__global__ void __launch_bounds__(512) wgrad_pc_kernel(const WgradP p) {
    constexpr int NT = 256;                      // threads per role
    constexpr int BM = WM * WVM * 32, BC = WN * WVN * 32;
    constexpr int PK = NPJ / 64;                 // pixel slots per producer lane
    constexpr int NPAIRS = (MODE == WG_SPATIAL) ? (R * XWE) / 2 : NPJ / 2;
    constexpr int CH = NPAIRS / WVK;             // pixel pairs per consumer wave and tile
    constexpr int HX = XWE / 2;                  // pixel pairs per tile row
    static_assert(WVM * WVN * WVK == 4 && NPJ % 64 == 0 && NPAIRS % WVK == 0, "4 consumer waves");
    static_assert(MODE != WG_SPATIAL || WVK == 1 || (HX % CH == 0) || (CH % HX == 0), "a wave's pairs: part of a row or whole rows");
    constexpr int WS = XWE + 2, IR = R + 2;      // halo of a 3x3 stride-1 tile
    constexpr int USED = (MODE == WG_SPATIAL) ? IR * WS : NPJ;     // Q elements (halo positions / pixels) per channel
    constexpr int NK = (USED + 63) / 64;         // 64-element groups of the Q image
    static_assert(MODE == WG_PIX || (T == 9 && R * XWE <= NPJ && XWE % 2 == 0), "3x3 tile geometry");
    static_assert(MODE == WG_SPATIAL || T == 1, "1x1: one tap");
    static_assert(MODE != WG_SPATIAL || PROP == S2K_PRO_NONE, "3x3 wgrad: P (= dY) carries no prologue");
    constexpr int BMP = BM + 4, BCP = BC + 4;    // row strides (floats): 16-byte aligned rows, stride = 4 mod 32 banks
    constexpr int NMQ = (BM + 15) / 16, NCQ = (BC + 15) / 16;  // channel quads per producer thread (a producer wave owns quads g, g+4, ...)
    static_assert(BM % 16 == 0 && BC % 16 == 0, "channel quads per producer wave");
    constexpr int PIMG = NPJ * BMP;
    constexpr int BUF = PIMG + USED * BCP;       // floats of one LDS image {P, Q}
    extern __shared__ __attribute__((aligned(16))) float smem[];

    WG_DBG_DECL();
    const bool producer = threadIdx.x >= NT;     // wave-uniform
    const int tid = threadIdx.x & (NT - 1);
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int mc = p.n_mtiles * p.n_ctiles;
    const int v = wg_xcd_remap(blockIdx.x, gridDim.x);
    const int split = v / mc, tl = v - split * mc;
    const int mt = tl % p.n_mtiles, ct = tl / p.n_mtiles;
    const int m0 = mt * BM, c0 = ct * BC;
    const int tile_begin = split * p.tiles_per_split;
    int tile_end = tile_begin + p.tiles_per_split;
    if (tile_end > p.ntiles) tile_end = p.ntiles;

    if (producer) {
        if (p.exp & 16) return;     // tuning builds: consumers alone (no staging wave, no barriers; results are garbage)
        // =================================================================================================================
        // PRODUCER wave g (0..3): pixel / halo slot = lane, channel quads g, g + 4, g + 8, ...
        // =================================================================================================================
        const int g = __builtin_amdgcn_readfirstlane(wave);
        const bool img_local = (MODE == WG_SPATIAL) || (p.HWp % NPJ) == 0;
        rsrc_t rp = make_rsrc(p.p, (int64_t)p.B * p.M * p.HWp * 4);
        rsrc_t rq = make_rsrc(p.q, (int64_t)p.B * p.C * p.HWq * 4);
        // scale / shift of this thread's channels stay in registers for the whole kernel
        float psc[NMQ][4], psh[NMQ][4], qsc[NCQ][4], qsh[NCQ][4];
#pragma unroll
        for (int i = 0; i < NMQ; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gm = min(m0 + (g + 4 * i) * 4 + q, p.M - 1);
                psc[i][q] = PROP != S2K_PRO_NONE ? p.bnvp[gm] : 1.0f;
                psh[i][q] = PROP != S2K_PRO_NONE ? p.bnvp[p.M + gm] : 0.0f;
            }
#pragma unroll
        for (int j = 0; j < NCQ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = min(c0 + (g + 4 * j) * 4 + q, p.C - 1);
                qsc[j][q] = PROQ != S2K_PRO_NONE ? p.bnvq[gc] : 1.0f;
                qsh[j][q] = PROQ != S2K_PRO_NONE ? p.bnvq[p.C + gc] : 0.0f;
            }
        float preg[PK][NMQ][4], qreg[NK][NCQ][4];
        unsigned f_pok = 0;          // bit k: pixel slot lane + 64 k is a real output pixel
        unsigned f_qok = 0;          // bit k: Q element lane + 64 k is inside the image (SPATIAL) / the pixel exists (PIX)
        const uint32_t p_rstep = (uint32_t)p.HWp * 4u, q_cstep = (uint32_t)p.HWq * 4u;

        auto fetch = [&](int tile) {
            uint32_t pvoff[PK], qvoff[NK];
            if (MODE == WG_SPATIAL) {
                const int tx = tile % p.tiles_x;
                const int ty = (tile / p.tiles_x) % p.tiles_y;
                const int b = tile / (p.tiles_x * p.tiles_y);
                const int y0 = ty * R, x0 = tx * p.XW;
                rp = make_rsrc(p.p + (int64_t)b * p.M * p.HWp, (int64_t)p.M * p.HWp * 4);
                rq = make_rsrc(p.q + (int64_t)b * p.C * p.HWq, (int64_t)p.C * p.HWq * 4);
                f_pok = 0;
#pragma unroll
                for (int k = 0; k < PK; ++k) {
                    const int pj = lane + 64 * k;
                    const int yo = y0 + pj / XWE, xo = x0 + pj % XWE;
                    const bool ok = pj < R * XWE && pj % XWE < p.XW && yo < p.HO && xo < p.WO;
                    pvoff[k] = ok ? (uint32_t)(yo * p.WO + xo) * 4u : BUF_OOB;
                    f_pok |= ok ? (1u << k) : 0u;
                }
                f_qok = 0;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int e = lane + 64 * k;
                    const int iy = y0 - p.PT + e / WS, ix = x0 - p.PL + e % WS;
                    const bool ok = e < USED && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    qvoff[k] = ok ? (uint32_t)(iy * p.W + ix) * 4u : BUF_OOB;
                    f_qok |= ok ? (1u << k) : 0u;
                }
            } else {
                static_assert(MODE == WG_SPATIAL || PK == 1, "1x1: 64 pixels per tile");
                const int64_t ntot = (int64_t)p.B * p.HWp;
                const int64_t n = (int64_t)tile * NPJ + lane;
                f_pok = n < ntot ? 1u : 0u;
                const int64_t nn = f_pok ? n : 0;
                const int b = (int)(nn / p.HWp);
                const int pp = (int)(nn - (int64_t)b * p.HWp);
                // descriptors based at the first image this pixel tile touches (uniform): offsets span the tile's images only
                const int bt = (int)(((int64_t)tile * NPJ) / p.HWp);
                rp = make_rsrc(p.p + (int64_t)bt * p.M * p.HWp, (int64_t)(img_local ? 1 : p.B - bt) * p.M * p.HWp * 4);
                rq = make_rsrc(p.q + (int64_t)bt * p.C * p.HWq, (int64_t)(img_local ? 1 : p.B - bt) * p.C * p.HWq * 4);
                const int brel = b - bt;
                pvoff[0] = f_pok ? (uint32_t)((int64_t)brel * p.M * p.HWp + pp) * 4u : BUF_OOB;
                qvoff[0] = f_pok ? (uint32_t)((int64_t)brel * p.C * p.HWq + pp) * 4u : BUF_OOB;
                f_qok = f_pok;
            }
            // rows past M / channels past C re-read the last valid one (they only feed discarded outputs); the row / channel part
            // of every address is a scalar offset
#pragma unroll
            for (int i = 0; i < NMQ; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t so = (uint32_t)min(m0 + (g + 4 * i) * 4 + q, p.M - 1) * p_rstep;
#pragma unroll
                    for (int k = 0; k < PK; ++k) preg[k][i][q] = bload_s(rp, pvoff[k], so);
                }
#pragma unroll
            for (int j = 0; j < NCQ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t so = (uint32_t)min(c0 + (g + 4 * j) * 4 + q, p.C - 1) * q_cstep;
#pragma unroll
                    for (int k = 0; k < NK; ++k) qreg[k][j][q] = bload_s(rq, qvoff[k], so);
                }
        };

        auto commit = [&](float* Pt, float* Qt) {
#pragma unroll
            for (int k = 0; k < PK; ++k) {
                const bool ok = (f_pok >> k) & 1u;
                const float bound = ok ? __builtin_inff() : 0.0f, keep = ok ? 1.0f : 0.0f;
                if (MODE != WG_SPATIAL || lane + 64 * k < R * XWE) {
#pragma unroll
                    for (int i = 0; i < NMQ; ++i) {
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = pro_masked<PROP>(preg[k][i][q], psc[i][q], psh[i][q], bound, keep);
                        *reinterpret_cast<f32x4*>(Pt + (lane + 64 * k) * BMP + (g + 4 * i) * 4) = o;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const bool ok = (f_qok >> k) & 1u;
                const float bound = ok ? __builtin_inff() : 0.0f, keep = ok ? 1.0f : 0.0f;
                if (lane + 64 * k < USED) {
#pragma unroll
                    for (int j = 0; j < NCQ; ++j) {
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = pro_masked<PROQ>(qreg[k][j][q], qsc[j][q], qsh[j][q], bound, keep);
                        *reinterpret_cast<f32x4*>(Qt + (lane + 64 * k) * BCP + (g + 4 * j) * 4) = o;
                    }
                }
            }
        };

        if (tile_begin < tile_end) fetch(tile_begin);
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            float* Pt = smem + ((tile - tile_begin) & 1) * BUF;
            const unsigned long long s0 = WG_STAMP();
            commit(Pt, Pt + PIMG);
            if (tile + 1 < tile_end && !(p.exp & 1)) fetch(tile + 1);     // in flight while the consumers work on this tile
            const unsigned long long s1 = WG_STAMP();
            __syncthreads();
            WG_DBG_ADD(4, s1 - s0);
            WG_DBG_ADD(3, WG_STAMP() - s1);
        }
        WG_DBG_FLUSH();
        return;
    }

    // =====================================================================================================================
    // CONSUMER: LDS reads + MFMAs, nothing else.
    // =====================================================================================================================
    __builtin_amdgcn_s_setprio(2);     // the MFMA stream wins issue arbitration against the staging wave of its SIMD
    const int wk = wave % WVK, wmn = wave / WVK;
    const int wm0 = (wmn / WVN) * (WM * 32), wc0 = (wmn % WVN) * (WN * 32);
    f32x16 acc[T][WM][WN];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.0f;

    // this wave's pairs: [wk * CH, (wk + 1) * CH) — a run inside one tile row, or whole rows
    const int s_base = wk * CH;
    const int qp_base = (MODE == WG_SPATIAL) ? (s_base / HX) * WS + 2 * (s_base % HX) : 2 * s_base;
    const int a_off = (lh + 2 * s_base) * BMP + wm0 + l31;               // + (2 s') * BMP + rm * 32
    const int b_off = PIMG + (lh + qp_base) * BCP + wc0 + l31;           // + (element of pair s' / tap) * BCP + rn * 32   (all compile-time)

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const unsigned long long s0 = WG_STAMP();
        if (!(p.exp & 16)) __syncthreads();                       // buffer (k & 1) is full
        const unsigned long long s1 = WG_STAMP();
        WG_DBG_ADD(0, s1 - s0);
        WG_DBG_ADD(2, 1);
        if (p.exp & 4) continue;
        const float* img = smem + ((tile - tile_begin) & 1) * BUF;
        const float* Pa = img + a_off;
        const float* Qb = img + b_off;
        auto lds_operands = [&](int s, float (&a)[WM], float (&bq)[T][WN]) {   // s is a compile-time constant after unrolling
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = Pa[2 * s * BMP + rm * 32];
            const int qp = (MODE == WG_SPATIAL && CH > HX) ? (s / HX) * WS + 2 * (s % HX) : 2 * s;   // relative to the wave's first pair
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int toff = (MODE == WG_SPATIAL) ? (t / 3) * WS + (t % 3) : 0;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) bq[t][rn] = Qb[(qp + toff) * BCP + rn * 32];
            }
        };
        auto mfmas = [&](const float (&a)[WM], const float (&bq)[T][WN]) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        acc[t][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm], bq[t][rn], acc[t][rm][rn], 0, 0, 0);
        };
        auto interleave = [&]() {      // MFMA, then a slice of the NEXT pair's LDS reads, MFMA, ...
#pragma unroll
            for (int t = 0; t < T * WM * WN; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // <= 2 DS reads
            }
        };
        float a0[WM], a1[WM], b0[T][WN], b1[T][WN];
        lds_operands(0, a0, b0);
#pragma unroll
        for (int s = 0; s < CH; s += 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < CH) lds_operands(s + 1, a1, b1);
            mfmas(a0, b0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < CH) {
                if (s + 2 < CH) lds_operands(s + 2, a0, b0);
                mfmas(a1, b1);
                interleave();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        WG_DBG_ADD(1, WG_STAMP() - s1);
    }

    // ---------------- combine: wgs[t][m][c] += acc (k-waves reduced through LDS first: wgrad.h) -------
    WG_DBG_FLUSH();
    if (p.exp & 2) return;
    if (tile_begin >= tile_end) return;
    if constexpr (WVK > 1) __syncthreads();     // every consumer wave has left the multiply loop (the producers have exited): the images are free
    wg_combine<T, WM, WN, WVK>(p, acc, smem, wk, wmn, lane, m0, c0, wm0, wc0);
}

// -------------------------------------------------------------------------------------------------
template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK, int NPJ, int R, int XWE, int PROP, int PROQ>
static int launch_pc2(WgradP& p, hipStream_t st) {
    constexpr int BM = WM * WVM * 32, BC = WN * WVN * 32;
    constexpr int USED = (MODE == WG_SPATIAL) ? (R + 2) * (XWE + 2) : NPJ;
    p.n_mtiles = cdiv(p.M, BM);
    p.n_ctiles = cdiv(p.C, BC);
    {   // 32-bit buffer offsets: one image must stay below 2 GiB when tiles are image-local, else the whole tensor
        const int64_t span = ((MODE == WG_SPATIAL) || (p.HWp % NPJ) == 0) ? 1 : std::min<int64_t>(p.B, (NPJ - 2) / p.HWp + 2);   // images one pixel tile touches
        const int64_t need = std::max((int64_t)p.M * p.HWp, (int64_t)p.C * p.HWq) * 4 * span;
        if (need >= 0x7ffffff0ll) { set_error("wgrad: the %lld image(s) one pixel tile touches exceed 2 GiB", (long long)span); return S2K_EINVAL; }
    }
    if (p.gatep || p.gateq) { set_error("wgrad (pc): SE gates stay on the generic kernels"); return S2K_EINVAL; }
    const size_t lds = (size_t)2 * (NPJ * (BM + 4) + USED * (BC + 4)) * sizeof(float);
    static_assert(2 * (NPJ * (BM + 4) + USED * (BC + 4)) * sizeof(float) <= 160 * 1024, "LDS image");
    if (lds > 160 * 1024) { set_error("wgrad: LDS %zu too large", lds); return S2K_EINVAL; }
    auto kern = wgrad_pc_kernel<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, R, XWE, PROP, PROQ>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // Pixel splits: one 8-wave workgroup per CU (LDS); pick the split count whose workgroup count fills whole rounds of the
    // 256 CUs, preferring fewer splits (each ends in an atomic combine of its accumulator tile).
    const int mc = p.n_mtiles * p.n_ctiles;
    const int slots = 256;
    int max_splits = cdiv(p.ntiles, 4);          // at least 4 pixel tiles per split
    if (max_splits > 65535) max_splits = 65535;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 1e30;
    const int s_hi = std::min(max_splits, std::max(1, 4 * slots / mc));
    for (int sp = 1; sp <= s_hi; ++sp) {
        const double rounds = (double)cdiv(mc * sp, slots);
        const double cost = rounds * ((double)cdiv(p.ntiles, sp) + (T == 9 ? 1.0 : (WM * WN >= 4 ? 2.0 : 1.0)));
        if (cost < best * 0.999) { best = cost; splits = sp; }
    }
    p.tiles_per_split = cdiv(p.ntiles, splits);
    splits = cdiv(p.ntiles, p.tiles_per_split);
    hipLaunchKernelGGL(kern, dim3(mc * splits), dim3(512), lds, st, p);
    g_s2k_variant = 1;
    return S2K_OK;
}

template <int R, int XWE>
static int launch_pc_spatial(WgradP& p, hipStream_t st) {
    if (p.proq == S2K_PRO_NONE) return launch_pc2<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 64, R, XWE, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
    if (p.proq == S2K_PRO_RELU) return launch_pc2<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 64, R, XWE, S2K_PRO_NONE, S2K_PRO_RELU>(p, st);
    return 1;
}

// thin layers (a 32-channel side): 128-pixel tiles (2 rows x 64), the four consumer waves split the pixel pairs
template <int WVM>
static int launch_pc_spatial_thin(WgradP& p, hipStream_t st) {
    if (p.proq == S2K_PRO_NONE) return launch_pc2<WG_SPATIAL, 9, 1, 1, WVM, 1, 4 / WVM, 128, 2, 64, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
    if (p.proq == S2K_PRO_RELU) return launch_pc2<WG_SPATIAL, 9, 1, 1, WVM, 1, 4 / WVM, 128, 2, 64, S2K_PRO_NONE, S2K_PRO_RELU>(p, st);
    return 1;
}

template <int WM, int WN>
static int launch_pc_pix(WgradP& p, hipStream_t st) {
    const int pp = p.prop, pq = p.proq;
    if (pp == S2K_PRO_NONE && pq == S2K_PRO_NONE) return launch_pc2<WG_PIX, 1, WM, WN, 2, 2, 1, 64, 1, 64, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
    if (pp == S2K_PRO_NONE && pq == S2K_PRO_RELU) return launch_pc2<WG_PIX, 1, WM, WN, 2, 2, 1, 64, 1, 64, S2K_PRO_NONE, S2K_PRO_RELU>(p, st);
    if (pp == S2K_PRO_RELU && pq == S2K_PRO_NONE) return launch_pc2<WG_PIX, 1, WM, WN, 2, 2, 1, 64, 1, 64, S2K_PRO_RELU, S2K_PRO_NONE>(p, st);
    return 1;      // SiLU / GELU prologues cost a producer too many instructions per tile: generic kernels
}

#if defined(S2K_TUNING) && defined(S2K_DMA_STAMPS)
extern "C" int s2k_debug_wg_counters(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_dbg), sizeof(g_wg_dbg)) != hipSuccess) return S2K_EHIP;
    if (reset) {
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_dbg), z, sizeof(z));
    }
    return S2K_OK;
}
#endif

// The shapes this file covers; everything else (thin layers, stride 2, 2x2 gather, SE-gated / SiLU operands, odd tilings)
// returns 1 and runs on the generic kernels of wgrad.hip.
int launch_wgrad_pc(WgradP& p, int mode, hipStream_t st) {
    static const int enabled = tune_int("S2K_WG_PC", 3);      // bit 0: 3x3, bit 1: 1x1
    const int64_t npix = (int64_t)p.B * p.HWp;
    if (p.gatep || p.gateq) return 1;
    if ((enabled & 1) && mode == S2K_MODE_CONV && p.T == 9 && p.S == 1 && p.KH == 3 && p.KW == 3 && p.H == p.HO && p.W == p.WO) {
        const bool thin = (p.M <= 32) || (p.M <= 64 && p.C <= 32);     // few channels on one side
        if (thin) {
            if (p.C > 32 || p.WO < 64 || p.HO < 2) return 1;           // (the 128-pixel kernels of wgrad.hip)
            p.R = 2; p.XW = 64; p.XWe = 64; p.IR = 4; p.IC = 66; p.WS = 66;
            p.tiles_x = cdiv(p.WO, 64);
            p.tiles_y = cdiv(p.HO, 2);
            p.ntiles = p.B * p.tiles_x * p.tiles_y;
            p.NP = 0;
            return p.M <= 32 ? launch_pc_spatial_thin<1>(p, st) : launch_pc_spatial_thin<2>(p, st);
        }
        const int XW = p.WO <= 64 ? p.WO : 64;
        const int XWe = (XW + 1) & ~1;
        int R = 64 / XWe;
        if (R > p.HO) R = p.HO;
        auto setup = [&](int r, int xwe) {
            p.R = r; p.XW = XW; p.XWe = xwe;
            p.IR = r + 2; p.IC = xwe + 2; p.WS = xwe + 2;
            p.tiles_x = cdiv(p.WO, XW);
            p.tiles_y = cdiv(p.HO, r);
            p.ntiles = p.B * p.tiles_x * p.tiles_y;
            p.NP = 0;
        };
#define PC_SPATIAL_CASE(RR, XX) \
        if (R == RR && XWe == XX) { setup(RR, XX); return launch_pc_spatial<RR, XX>(p, st); }
        PC_SPATIAL_CASE(1, 64)
        PC_SPATIAL_CASE(2, 32)
        PC_SPATIAL_CASE(4, 16)
        PC_SPATIAL_CASE(8, 8)
        PC_SPATIAL_CASE(1, 56)
        PC_SPATIAL_CASE(2, 28)
        PC_SPATIAL_CASE(4, 14)
#undef PC_SPATIAL_CASE
        return 1;
    }
    if ((enabled & 2) && mode == S2K_MODE_CONV && p.T == 1 && p.S == 1 && p.H == p.HO && p.W == p.WO) {
        if (p.M <= 32 || p.C <= 32) return 1;                           // thin: generic kernels (high occupancy already)
        if (npix < 1024) return 1;
        p.NP = 64;
        p.ntiles = (int)cdiv64(npix, 64);
        // tile edge per side: 128 unless it pads the side by more than 12 % (176 -> 256 wastes 45 %, 3 x 64 = 192 wastes 9 %)
        auto edge = [](int n) { return (n > 64 && (double)cdiv(n, 128) * 128 / n <= 1.12) ? 128 : 64; };
        const int em = edge(p.M), ec = edge(p.C);
        if (em == 128 && ec == 128) return launch_pc_pix<2, 2>(p, st);
        if (em == 128) return launch_pc_pix<2, 1>(p, st);
        if (ec == 128) return launch_pc_pix<1, 2>(p, st);
        return launch_pc_pix<1, 1>(p, st);
    }
    return 1;
}

}  // namespace s2k
