// Weight gradients of the 1x1 convs / Linears, producer / consumer form with QUAD operand reads (round 4).
// Same arithmetic and reference code as wgrad_pc.hip's pixel mode (dW[m][c] += sum_pix Ppro[m][pix] * Qpro[c][pix]; ATen
// convolution_backward grad_weight for efficientnet_unet.py:319-372 and every Linear of timm's Block, prithvi.py:162-183), same roles
// (waves 0-3 consume: LDS reads + MFMAs; waves 4-7 stage the next 64-pixel tile; one barrier per tile), same combine.  What changes is
// the operand traffic of the consumers, which is what bounds them (DESIGN.md lesson 19: every LDS read beside the f32 MFMA stream costs
// 6 - 9 matrix-pipe cycles; the 2 x 2 wave tile of wgrad_pc_kernel issues one read per MFMA = 73.6 cycles per 64-cycle MFMA):
//
//   * The contraction runs over PIXELS, and a sum does not care in which order its terms arrive: k-step s' of pixel group g takes, in
//     lane half lh, pixel 8 g + 4 lh + s' instead of 2 s + lh.  The four k-steps of a group then want four CONSECUTIVE pixels of the
//     lane's channel - with a channel-major image [channel][pixel] (row stride 64 + 4 floats: 8 consecutive rows cover all 32 banks)
//     that is ONE ds_read_b128 per operand tile and group: (WM + WN) reads per 4 WM WN MFMAs, 0.25 per MFMA for the 2 x 2 tile.
//   * Channel-major is also the layout of the tensors in HBM, so a producer thread moves 16 bytes HBM -> registers -> (BatchNorm + ReLU)
//     -> LDS with one buffer_load_dwordx4 and one ds_write_b128 per four elements (wgrad_pc_kernel: four dword loads per write), i.e.
//     a quarter of the load instructions beside the MFMA stream that starves its SIMD partner.
//
// Needs H * W % 4 == 0 (16-byte pixel quads inside one image); everything else stays on wgrad_pc.hip / wgrad.hip.
#include <algorithm>

#include "common.h"
#include "wgrad.h"

namespace s2k {

template <int PRO>
__device__ __forceinline__ float q4_pro(float v, float sc, float sh, float bound) {
    if (PRO == S2K_PRO_NONE) return v;                                   // out-of-range pixels were loaded as 0
    return __builtin_amdgcn_fmed3f(fmaf(v, sc, sh), 0.0f, bound);      // ReLU; bound = 0 for a pixel past the end (else +inf)
}

// BatchNorm + SiLU + SE gate of four pixels of one channel (the MBConv project conv's operand): two-wide vector arithmetic so that the
// affine part, the 1 + e^-u and the final products are v_pk_* instructions - the producer's budget beside the MFMA stream is its
// INSTRUCTION count.  `gate` is 0 for a quad past the end.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 q4_silu_gate(f32x4 v, float sc, float sh, float gate) {
#ifdef S2K_EXACT_SILU
    return f32x4{silu_f(fmaf(v[0], sc, sh)) * gate, silu_f(fmaf(v[1], sc, sh)) * gate, silu_f(fmaf(v[2], sc, sh)) * gate, silu_f(fmaf(v[3], sc, sh)) * gate};
#endif
    const f32x2 s2 = {sc, sc}, h2 = {sh, sh}, g2 = {gate, gate}, one = {1.0f, 1.0f};
    f32x2 u0 = {v[0], v[1]}, u1 = {v[2], v[3]};
    u0 = u0 * s2 + h2;
    u1 = u1 * s2 + h2;
    f32x2 e0 = {__expf(-u0[0]), __expf(-u0[1])}, e1 = {__expf(-u1[0]), __expf(-u1[1])};
    e0 += one;
    e1 += one;
    const f32x2 r0 = {__builtin_amdgcn_rcpf(e0[0]), __builtin_amdgcn_rcpf(e0[1])}, r1 = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
    u0 = u0 * g2 * r0;
    u1 = u1 * g2 * r1;
    return f32x4{u0[0], u0[1], u1[0], u1[1]};
}

template <int WM, int WN, int PROP, int PROQ>
__global__ void __launch_bounds__(512) wgrad_q4_kernel(const WgradP p) {
    constexpr int NT = 256;
    constexpr int BM = WM * 64, BC = WN * 64;      // consumer waves 2 x 2, wave tile (32 WM) x (32 WN)
    constexpr int NPJ = 64;                        // pixels per tile
    constexpr int PS = NPJ + 4;                    // row stride (floats): 16-byte aligned rows, stride = 4 mod 32 banks
    constexpr int NG = NPJ / 8;                    // pixel groups (4 k-steps each) per tile
    constexpr int NMP = BM / 16, NCP = BC / 16;    // rows per producer thread and tile (16 rows per pass: 16 quads x 16 rows = 256 threads)
    constexpr int PIMG = BM * PS, BUF = (BM + BC) * PS;
    static_assert(PROP == S2K_PRO_NONE || PROP == S2K_PRO_RELU, "P prologue");
    static_assert(PROQ == S2K_PRO_NONE || PROQ == S2K_PRO_RELU || PROQ == S2K_PRO_SILU, "Q prologue (SiLU: with the SE gate)");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool producer = threadIdx.x >= NT;       // wave-uniform
    const int tid = threadIdx.x & (NT - 1);
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int mc = p.n_mtiles * p.n_ctiles;
    const int v = wg_xcd_remap(blockIdx.x, gridDim.x);
    // This workgroup's work: units [u0, u1) of the list of (output tile tl, 64-pixel tile) pairs, output-tile major.  Split form: the
    // pixel tiles [split * tiles_per_split, ...) of ONE output tile.  Stream form (p.streamk: the launcher picks it where whole rounds
    // of 256 workgroups fit badly - 144 output tiles of 52 pixel tiles cost 3 rounds x 13 units split five ways, 29.25 + 2 flushes
    // here): the list is cut into gridDim.x equal ranges, which may cross into the next output tile(s); the accumulators are
    // flushed (float atomics, as ever) at every crossing.
    int u0, u1;
    if (p.streamk) {
        const int64_t total = (int64_t)mc * p.ntiles;
        u0 = (int)(total * v / gridDim.x);
        u1 = (int)(total * (v + 1) / gridDim.x);
    } else {
        const int split = v / mc, tl = v - split * mc;
        const int tb = split * p.tiles_per_split;
        u0 = tl * p.ntiles + tb;
        u1 = tl * p.ntiles + min(tb + p.tiles_per_split, p.ntiles);
    }
    if (u0 >= u1) return;                          // (whole workgroup)

    if (producer) {
        // =================================================================================================================
        // PRODUCER thread: pixel quad q = tid & 15 of rows r + 16 i (r = tid >> 4) of both images
        // =================================================================================================================
        const int q = tid & 15, r = tid >> 4;
        const uint32_t ntot = (uint32_t)p.B * (uint32_t)p.HWp;      // < 2^31 (launcher): 32-bit index arithmetic throughout - a 64-bit
                                                                     // division is ~150 instructions, and a producer gets about one per MFMA
        const bool img_local = (p.HWp % NPJ) == 0;
        float psc[NMP], psh[NMP], qsc[NCP], qsh[NCP];
        uint32_t prow[NMP], qrow[NCP];              // byte offset of the (clamped) row inside an image
        int qch[NCP];                               // the (clamped) channel: index of its SE gate
        float qgate[NCP];                           // SiLU: gate[b][channel] of the tile held in qreg (0 for a quad past the end)
        auto set_rows = [&](int tl) {               // the output tile's rows: once per segment
            const int m0 = (tl % p.n_mtiles) * BM, c0 = (tl / p.n_mtiles) * BC;
#pragma unroll
            for (int i = 0; i < NMP; ++i) {
                const int gm = min(m0 + r + 16 * i, p.M - 1);      // rows past M re-read the last one (they feed discarded outputs)
                prow[i] = (uint32_t)gm * (uint32_t)p.HWp * 4u;
                psc[i] = PROP != S2K_PRO_NONE ? p.bnvp[gm] : 1.0f;
                psh[i] = PROP != S2K_PRO_NONE ? p.bnvp[p.M + gm] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < NCP; ++j) {
                const int gc = min(c0 + r + 16 * j, p.C - 1);
                qrow[j] = (uint32_t)gc * (uint32_t)p.HWq * 4u;
                qch[j] = gc;
                qsc[j] = PROQ != S2K_PRO_NONE ? p.bnvq[gc] : 1.0f;
                qsh[j] = PROQ != S2K_PRO_NONE ? p.bnvq[p.C + gc] : 0.0f;
            }
        };
        f32x4 preg[NMP], qreg[NCP];
        float bound = 0.0f;                          // of the tile held in preg / qreg
        auto fetch = [&](int tile) {
            const uint32_t n = (uint32_t)tile * NPJ + 4u * q;
            const bool ok = n < ntot;                // HWp % 4 == 0: a quad is inside one image and entirely valid or not
            const uint32_t nn = ok ? n : 0u;
            const uint32_t b = nn / (uint32_t)p.HWp;
            const uint32_t pp = nn - b * (uint32_t)p.HWp;
            // descriptors based at the first image this pixel tile touches (uniform): offsets span the tile's images only
            const uint32_t bt = __builtin_amdgcn_readfirstlane(((uint32_t)tile * NPJ) / (uint32_t)p.HWp);
            const rsrc_t rp = make_rsrc(p.p + (int64_t)bt * p.M * p.HWp, (int64_t)(img_local ? 1 : p.B - (int)bt) * p.M * p.HWp * 4);
            const rsrc_t rq = make_rsrc(p.q + (int64_t)bt * p.C * p.HWq, (int64_t)(img_local ? 1 : p.B - (int)bt) * p.C * p.HWq * 4);
            const uint32_t pv = ok ? ((b - bt) * (uint32_t)p.M * (uint32_t)p.HWp + pp) * 4u : BUF_OOB;       // (< 2 GiB: the launcher's span check)
            const uint32_t qv = ok ? ((b - bt) * (uint32_t)p.C * (uint32_t)p.HWq + pp) * 4u : BUF_OOB;
#pragma unroll
            for (int i = 0; i < NMP; ++i) preg[i] = bload4(rp, pv + prow[i]);     // BUF_OOB + a row offset (< 2^31) stays out of range: no select,
                                                                              // which hipcc would turn into a branch and a full wait per load
#pragma unroll
            for (int j = 0; j < NCP; ++j) qreg[j] = bload4(rq, qv + qrow[j]);
            if (PROQ == S2K_PRO_SILU) {
#pragma unroll
                for (int j = 0; j < NCP; ++j) {
                    const float gv = p.gateq ? p.gateq[(int64_t)b * p.C + qch[j]] : 1.0f;     // (b = 0 for a quad past the end: a valid address)
                    qgate[j] = ok ? gv : 0.0f;
                }
            }
            bound = ok ? __builtin_inff() : 0.0f;
        };
        auto commit = [&](float* Pt, float* Qt) {
#pragma unroll
            for (int i = 0; i < NMP; ++i) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = q4_pro<PROP>(preg[i][e], psc[i], psh[i], bound);
                *reinterpret_cast<f32x4*>(Pt + (r + 16 * i) * PS + 4 * q) = o;
            }
#pragma unroll
            for (int j = 0; j < NCP; ++j) {
                f32x4 o;
                if (PROQ == S2K_PRO_SILU) o = q4_silu_gate(qreg[j], qsc[j], qsh[j], qgate[j]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = q4_pro<(PROQ == S2K_PRO_SILU ? S2K_PRO_NONE : PROQ)>(qreg[j][e], qsc[j], qsh[j], bound);
                }
                *reinterpret_cast<f32x4*>(Qt + (r + 16 * j) * PS + 4 * q) = o;
            }
        };
        int tl = u0 / p.ntiles, tile = u0 - tl * p.ntiles;
        set_rows(tl);
        fetch(tile);
        for (int u = u0, it = 0; u < u1; ++it) {
            float* Pt = smem + (it & 1) * BUF;
            commit(Pt, Pt + PIMG);
            ++u;
            if (u < u1 && !(p.exp & 1)) {              // the next unit is in flight while the consumers work on this one
                if (++tile == p.ntiles) { tile = 0; ++tl; set_rows(tl); }     // (tuning builds, S2K_WG_EXP: 1 = stage the first unit only,
                fetch(tile);                                                    //  4 = no MFMA loop)
            }
            __syncthreads();
        }
        return;
    }

    // =====================================================================================================================
    // CONSUMER: LDS reads + MFMAs, nothing else.
    // =====================================================================================================================
    __builtin_amdgcn_s_setprio(2);
    const int wm0 = (wave >> 1) * (WM * 32), wc0 = (wave & 1) * (WN * 32);
    f32x16 acc[1][WM][WN];
    const int a_off = (wm0 + l31) * PS + 4 * lh;                 // + rm * 32 * PS + 8 g
    const int b_off = PIMG + (wc0 + l31) * PS + 4 * lh;          // + rn * 32 * PS + 8 g

    int it = 0;
    for (int u = u0; u < u1;) {
      const int tl = u / p.ntiles;
      const int seg_end = min(u1, (tl + 1) * p.ntiles);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) acc[0][i][j][rr] = 0.0f;
      for (; u < seg_end; ++u, ++it) {
        __syncthreads();                                         // buffer (it & 1) is full
        if (p.exp & 4) continue;
        const float* img = smem + (it & 1) * BUF;
        const float* Pa = img + a_off;
        const float* Qb = img + b_off;
        auto lds_operands = [&](int g, f32x4 (&a)[WM], f32x4 (&b)[WN]) {      // g is a compile-time constant after unrolling
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = *reinterpret_cast<const f32x4*>(Pa + rm * 32 * PS + 8 * g);
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) b[rn] = *reinterpret_cast<const f32x4*>(Qb + rn * 32 * PS + 8 * g);
        };
        auto mfmas = [&](const f32x4 (&a)[WM], const f32x4 (&b)[WN]) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn)
                        acc[0][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm][s], b[rn][s], acc[0][rm][rn], 0, 0, 0);
        };
        auto interleave = [&]() {      // the next group's (WM + WN) reads spread over this group's MFMAs
#pragma unroll
            for (int t = 0; t < WM + WN; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, (4 * WM * WN) / (WM + WN), 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        f32x4 a0[WM], a1[WM], b0[WN], b1[WN];
        lds_operands(0, a0, b0);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            __builtin_amdgcn_sched_barrier(0);
            lds_operands(g + 1, a1, b1);
            mfmas(a0, b0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) lds_operands(g + 2, a0, b0);
            mfmas(a1, b1);
            interleave();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // flush this output tile's sums (no LDS involved: the producers may already be staging the next segment's first unit)
      wg_combine<1, WM, WN, 1>(p, acc, smem, 0, wave, lane, (tl % p.n_mtiles) * BM, (tl / p.n_mtiles) * BC, wm0, wc0);
    }
}

template <int WM, int WN, int PROP, int PROQ>
static int launch_q4w(WgradP& p, hipStream_t st) {
    constexpr int BM = WM * 64, BC = WN * 64, NPJ = 64;
    constexpr size_t lds = (size_t)2 * (BM + BC) * (NPJ + 4) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS image");
    p.n_mtiles = cdiv(p.M, BM);
    p.n_ctiles = cdiv(p.C, BC);
    p.NP = NPJ;
    p.ntiles = (int)cdiv64((int64_t)p.B * p.HWp, NPJ);
    {   // 32-bit buffer offsets: see launch_pc2 (wgrad_pc.hip)
        const int64_t span = (p.HWp % NPJ) == 0 ? 1 : std::min<int64_t>(p.B, (NPJ - 2) / p.HWp + 2);
        const int64_t need = std::max((int64_t)p.M * p.HWp, (int64_t)p.C * p.HWq) * 4 * span;
        if (need >= 0x7ffffff0ll) { set_error("wgrad: the %lld image(s) one pixel tile touches exceed 2 GiB", (long long)span); return S2K_EINVAL; }
    }
    auto kern = wgrad_q4_kernel<WM, WN, PROP, PROQ>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // pixel splits as in launch_pc2: the split count whose workgroup count fills whole rounds of the 256 CUs, preferring fewer splits
    const int mc = p.n_mtiles * p.n_ctiles;
    const int slots = 256;
    int max_splits = std::max(1, std::min(65535, cdiv(p.ntiles, 4)));
    int splits = 1;
    double best = 1e30;
    const int s_hi = std::min(max_splits, std::max(1, 4 * slots / mc));
    for (int sp = 1; sp <= s_hi; ++sp) {
        const double rounds = (double)cdiv(mc * sp, slots);
        const double cost = rounds * ((double)cdiv(p.ntiles, sp) + (WM * WN >= 4 ? 2.0 : 1.0));
        if (cost < best * 0.999) { best = cost; splits = sp; }
    }
    p.tiles_per_split = cdiv(p.ntiles, splits);
    splits = cdiv(p.ntiles, p.tiles_per_split);
    // stream form: equal unit ranges over one round of workgroups; a range of n units crosses into 1 + (n - 1) / ntiles further output
    // tiles at most, and every crossing (and the end) is a flush.  OFF in the shipped library: alone it is what the model says
    // (768 x 3072 over 3,328 tokens 87 -> 105 TF/s, 2304 x 768 88 -> 102; all weight gradients of the MAE step 13.2 -> 12.0 ms), inside
    // the step it LOSES (alternating runs, tools/exp_wq4_step.sh: 41.59 vs 41.42 ms) - these kernels run on the side stream beside the
    // data gradients of the critical path, and 256 workgroups that hold their CU for the whole kernel leave those no gaps, whereas
    // the pixel splits' short workgroups (and the idle CUs of their last round) do.  DESIGN.md lesson 22.
    static const int streamk = tune_int("S2K_WG_STREAMK", 0);      // 0: never, 1: where the cost model prefers it, 2: always
    const int64_t total = (int64_t)mc * p.ntiles;
    const int grid_sk = (int)std::min<int64_t>(slots, total);
    const double per_wg = (double)cdiv64(total, grid_sk);
    const double cost_sk = per_wg + (WM * WN >= 4 ? 2.0 : 1.0) * (2.0 + (double)((int64_t)(per_wg - 1) / p.ntiles));
    p.streamk = (streamk == 2 || (streamk == 1 && cost_sk < 0.95 * best)) ? 1 : 0;
    if (total > 0x7fffffff) p.streamk = 0;
    hipLaunchKernelGGL(kern, dim3(p.streamk ? grid_sk : mc * splits), dim3(512), lds, st, p);
    g_s2k_variant = 4;
    return S2K_OK;
}

template <int WM, int WN>
static int launch_q4w_pro(WgradP& p, hipStream_t st) {
    const int pp = p.prop, pq = p.proq;
    if (pp == S2K_PRO_NONE && pq == S2K_PRO_NONE) return launch_q4w<WM, WN, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
    if (pp == S2K_PRO_NONE && pq == S2K_PRO_RELU) return launch_q4w<WM, WN, S2K_PRO_NONE, S2K_PRO_RELU>(p, st);
    if (pp == S2K_PRO_RELU && pq == S2K_PRO_NONE) return launch_q4w<WM, WN, S2K_PRO_RELU, S2K_PRO_NONE>(p, st);
    if (pp == S2K_PRO_NONE && pq == S2K_PRO_SILU) return launch_q4w<WM, WN, S2K_PRO_NONE, S2K_PRO_SILU>(p, st);      // (+ the SE gate of Q)
    return 1;
}

// (A 3x3 form of this kernel existed in round 4 - `wgrad_q4s_kernel`: channel-major halo tile [c][R + 2][XW + 8], an aligned 16-byte group
// plus the two dwords beside it serving the three taps of a row, 10 LDS reads per 36 MFMAs instead of 10 per 9.  Alone it ran the decoder's
// weight gradients at 118 - 124 TF/s against wgrad_pc_kernel's 112 - 118; in the step it changed nothing (alternating runs, tools/
// exp_wq4_unet.sh: 32.57 vs 32.58 ms) - those stages sit on the side stream under the encoder's backward, off the critical path - and was
// removed.  git: "wgrad_q4s_kernel".)

// S2K_OK = launched, 1 = not one of this kernel's shapes (the caller goes on to wgrad_pc.hip / the generic kernels), < 0 = error
int launch_wgrad_q4(WgradP& p, int mode, hipStream_t st) {
    static const int enabled = tune_int("S2K_WG_Q4", 5);      // bit 0: 1x1, bit 2: 1x1 with the SiLU + SE-gate operand
    if (!enabled || p.gatep || p.p_bf16) return 1;
    if (p.gateq && (p.proq != S2K_PRO_SILU || p.T != 1 || !(enabled & 4))) return 1;      // the SE-gated operand of the project convs: 1x1, SiLU
    if (mode != S2K_MODE_CONV || p.S != 1 || p.H != p.HO || p.W != p.WO) return 1;
    if ((reinterpret_cast<uintptr_t>(p.p) | reinterpret_cast<uintptr_t>(p.q)) & 15) return 1;
    if (p.T != 1 || !(enabled & 1)) return 1;
    if (p.M <= 32 || p.C <= 32 || (p.HWp & 3) || p.HWq != p.HWp) return 1;
    if ((int64_t)p.B * p.HWp < 1024 || (int64_t)p.B * p.HWp > 0x7fffffffll) return 1;
    if ((reinterpret_cast<uintptr_t>(p.p) | reinterpret_cast<uintptr_t>(p.q)) & 15) return 1;
    // tile edge per side: 128 unless it pads the side by more than 12 % (as wgrad_pc.hip)
    auto edge = [](int n) { return (n > 64 && (double)cdiv(n, 128) * 128 / n <= 1.12) ? 128 : 64; };
    const int em = edge(p.M), ec = edge(p.C);
    if (em == 128 && ec == 128) return launch_q4w_pro<2, 2>(p, st);
    if (em == 128) return launch_q4w_pro<2, 1>(p, st);
    if (ec == 128) return launch_q4w_pro<1, 2>(p, st);
    return launch_q4w_pro<1, 1>(p, st);
}

}  // namespace s2k
