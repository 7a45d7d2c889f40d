// 1x1 convolution / Linear on the f32 matrix cores for inputs WITHOUT a load prologue: "quad" operand layout (round 4).
//
// Replaces what conv_dma.hip / conv_igemm_kernel / conv_pc_kernel do for the prologue-free 1x1 contractions of the hot path (MBConv
// expand convs efficientnet_unet.py:319-336, the encoder's 1x1 data gradients, the ViT Linears of prithvi.py:162-183 and their
// data gradients).  Same arithmetic (exact f32 MFMA, k ascending), same epilogue semantics (bias, residual, accumulate, BatchNorm
// batch statistics, split-K partials for splitk_reduce_kernel).
//
// What the measurements of this round say (tools/exp_mfma_kstep.py, profiles/r04_mfma_kstep.txt): v_mfma_f32_32x32x2_f32 issues
// every 64 cycles from registers whatever the number of accumulators, but EVERY other instruction of the wave - a ds_read_b32
// of an operand, a VALU op - adds 6 - 9 cycles to the matrix pipe's time (f32 MFMAs do not overlap the vector / LDS issue).
// A wave tile of WM x WN accumulators needs (WM + WN) operand dwords per WM * WN MFMAs: 2 x 2 tiles (conv_pc_kernel) run at
// 73.6 cycles per MFMA, 3 x 1 at 72.7, 1 x 1 (the generic kernel's 64 x 64 tiles, the narrow producer / consumer tiles) at 152.
// The lever is operand dwords PER INSTRUCTION:
//   * B (activations [k][pixels] in LDS, as they are in memory): a lane reads FOUR consecutive pixels of one channel with one
//     ds_read_b128 and uses them as the B operands of four pixel tiles - tile rn's column j is pixel 4 j + rn.  One read feeds
//     4 * WM MFMAs.
//   * A (weights): WEIGHT_PACK writes a second copy [k / 8][m][8] in which the eight channels of a group stand in the order
//     (k & 1) * 4 + (k >> 1): a lane's A operands of FOUR consecutive k-steps are 16 contiguous bytes = one ds_read_b128.
//   -> per 4 k-steps a wave issues WM + 4 reads for 16 * WM MFMAs (WM = 2: 0.19 reads per MFMA instead of 1.0).
//   * the accumulators of the four pixel tiles hold four CONSECUTIVE pixels of a row per lane: the epilogue stores 16-byte pixel
//     quads straight from the registers (2 rows x 512 bytes per instruction) - no LDS transposition, no barrier.
// Staging, ring, persistence and K split are conv_dma.hip's (LDS-DMA, `buffer_load_dwordx4 ... lds`): both images are lane-linear
// copies of memory.  Roles are split as in conv_pc_kernel, because a DMA piece costs its issuing wave ~55 cycles (the MFMAs of a
// 128 x 128 tile's 16-channel stage are 2,048): waves 0-3 are CONSUMERS (LDS reads, MFMAs, the epilogue's stores), waves 4-7
// PRODUCERS that do nothing but issue the pieces of stage g + NST - 1 and wait (counted vmcnt: they issue no stores, so the count is
// exact) for those of stage g + 1; one workgroup barrier per stage.  A producer wave shares its SIMD with a consumer and costs it
// ~7 cycles per instruction (tools/exp_mfma_kstep.py): 4 - 9 pieces per stage.
#include <algorithm>

#include "common.h"
#define DMA_DBG_SYM g_q4_dbg
#include "dma.h"
#include "igemm.h"

namespace s2k {

// wave tile: 32 * WM rows x 128 pixels; workgroup: WVM x WVN waves (4 in all) = 32 * WM * WVM rows x 128 * WVN pixels
// PRE: 0 = plain store; 1 = + bias (prefetched at the item's start); 2 = + bias, residual and / or old value (accumulate)
template <int WM, int WVM, int WVN, int NST, int PRE, int KCH>
__global__ void __launch_bounds__(512) conv_q4_kernel(const ConvP p) {
    static_assert(KCH == 16 || KCH == 32, "stage depth");
    constexpr int BM = 32 * WM * WVM, BN = 128 * WVN;
    constexpr int A_FL = KCH * BM, B_FL = KCH * BN, ST_FL = A_FL + B_FL;
    constexpr int PA = A_FL / 1024, PB = B_FL / 1024, PW = PA + PB;        // 1-KiB pieces per wave per stage
    static_assert(WVM * WVN == 4 && A_FL % 1024 == 0 && B_FL % 1024 == 0, "tile vs DMA pieces");
    static_assert(NST >= 3 && PW * (NST - 1) < 60, "ring depth vs the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    DMA_DBG_DECL();
    const unsigned long long t_begin = DMA_STAMP();
    const bool producer = threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..3 within the role
    const int wm0 = (wave / WVN) * (WM * 32), wn0 = (wave % WVN) * 128;
    const int HW = p.HW;
    const int nchunks = (p.Ctot + KCH - 1) / KCH;
    const int n_items = p.n_tiles;                       // item = (nt * splits + ks) * n_mtiles + mt
    const int pos = xcd_remap(blockIdx.x, gridDim.x);

    if (producer) {
    // =================================================================================================================================
    // PRODUCER: the stage stream of this workgroup (all its items back to back), NST - 1 stages ahead of the consumers
    // =================================================================================================================================
    // ---- this wave's DMA pieces: piece q = wave + 4 j covers stage floats [256 q, 256 q + 256) ------------------------------------
    // A image [kg][BM][8] <- mirror [(c0 / 8 + kg)][MP][8]: a piece = 32 rows of one k-group, contiguous on both sides
    // B image [k][BN]     <- x1 [c0 + k][pixels]: a piece = 256 pixels of one or two channel rows
    uint32_t a_fix[PA];
    int b_row[PB], b_col[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int f = 256 * (wave + 4 * j) + 4 * lane;
        const int kg = f / (BM * 8), r = f % (BM * 8);
        a_fix[j] = (uint32_t)(kg * p.w_st * 8 + r) * 4u;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int f = 256 * (wave + 4 * j) + 4 * lane;
        b_row[j] = f / BN;
        b_col[j] = f % BN;
    }
    const rsrc_t rw = make_rsrc(p.wtq, 0x7ffffff0ll);    // the mirror of a WPACK entry is zero padded to [KP / 8][MP][8]
    const uint32_t hw4 = (uint32_t)HW * 4u;

    // ---- producer state (conv_dma.hip) ---------------------------------------------------------------------------------------------
    int p_item = pos, p_st = 0, p_ns = 0, p_c0 = 0;
    uint32_t p_avoff[PA], p_bpix[PB];
    rsrc_t p_rx = make_rsrc(p.x1, 0);
    auto item_range = [&](int item, int& mt, int& nt, int& cb, int& ce) {
        const int r = item / p.n_mtiles;
        mt = __builtin_amdgcn_readfirstlane(item - r * p.n_mtiles);
        nt = __builtin_amdgcn_readfirstlane(r / p.splits);
        const int ks = r - nt * p.splits;
        cb = __builtin_amdgcn_readfirstlane((ks * nchunks) / p.splits);
        ce = __builtin_amdgcn_readfirstlane(((ks + 1) * nchunks) / p.splits);
    };
    auto producer_enter = [&]() {
        int mt, nt, cb, ce;
        item_range(p_item, mt, nt, cb, ce);
        p_ns = ce - cb;
        p_c0 = cb * KCH;
        p_st = 0;
        const int n0 = nt * BN;
        const int img_b = __builtin_amdgcn_readfirstlane(n0 / HW);
        p_rx = make_rsrc(p.x1 + (int64_t)img_b * p.C1 * HW, (int64_t)(p.B - img_b) * p.C1 * HW * 4);
#pragma unroll
        for (int j = 0; j < PA; ++j) p_avoff[j] = a_fix[j] + (uint32_t)(mt * BM * 8) * 4u;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int n = n0 + b_col[j];
            const int b = n / HW, pp = n - b * HW;
            p_bpix[j] = n < p.Ntot ? (uint32_t)((int64_t)(b - img_b) * p.C1 * HW + pp) * 4u : BUF_OOB;
        }
    };
    int issued = 0;
    bool p_done = false;
    auto issue = [&]() {
        float* slot = smem + __builtin_amdgcn_readfirstlane(issued % NST) * ST_FL;
        const int c0 = __builtin_amdgcn_readfirstlane(p_c0 + p_st * KCH);
        const uint32_t soa = (uint32_t)(c0 >> 3) * (uint32_t)p.w_st * 32u;      // k-group row of the mirror: MP * 8 floats
        const uint32_t sob = (uint32_t)c0 * hw4;
        const int cmax = p.Ctot - 1 - c0;
#pragma unroll
        for (int j = 0; j < PA; ++j) dma16(rw, slot + 256 * (wave + 4 * j), p_avoff[j], soa);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const uint32_t voff = p_bpix[j] + (uint32_t)min(b_row[j], cmax) * hw4;
            dma16(p_rx, slot + A_FL + 256 * (wave + 4 * j), voff, sob);
        }
        ++issued;
        if (++p_st == p_ns) {
            p_item += gridDim.x;
            if (p_item < n_items) producer_enter();
            else p_done = true;
        }
    };
    producer_enter();                    // (the launcher never starts more workgroups than items)
#pragma unroll 1
    for (int i = 0; i < NST - 1 && !p_done; ++i) issue();
    // total number of stages = what the consumers will wait for: one barrier per stage, both roles
    int total = 0;
#pragma unroll 1
    for (int item = pos; item < n_items; item += gridDim.x) {
        int mt, nt, cb, ce;
        item_range(item, mt, nt, cb, ce);
        total += ce - cb;
    }
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        // my pieces of stage g have landed once at most (issued - g - 1) younger stages are outstanding (loads only: exact)
        if (issued - g - 1 >= NST - 2) wait_vm<PW * (NST - 2)>();
        else wait_vm<0>();
        wg_barrier();                    // stage g is in LDS for everyone; the consumers have left slot (g - 1) % NST
        if (!p_done) issue();
    }
    return;
    }

    // ---- consumer ------------------------------------------------------------------------------------------------------------------
    const int a_lane = (wm0 + l31) * 8 + lh * 4;             // + (kg * BM + rm * 32) * 8: the lane's four k-steps of group kg
    const int b_lane = A_FL + lh * BN + wn0 + 4 * l31;       // + (2 s) * BN: the lane's four pixels (= four tiles) of k-step s
    // =================================================================================================================================
    // CONSUMER
    // =================================================================================================================================
    auto item_range = [&](int item, int& mt, int& nt, int& cb, int& ce) {
        const int r = item / p.n_mtiles;
        mt = __builtin_amdgcn_readfirstlane(item - r * p.n_mtiles);
        nt = __builtin_amdgcn_readfirstlane(r / p.splits);
        const int ks = r - nt * p.splits;
        cb = __builtin_amdgcn_readfirstlane((ks * nchunks) / p.splits);
        ce = __builtin_amdgcn_readfirstlane(((ks + 1) * nchunks) / p.splits);
    };
    int g = 0;
#pragma unroll 1
    for (int item = pos; item < n_items; item += gridDim.x) {
        int mt, nt, cb, ce;
        item_range(item, mt, nt, cb, ce);
        const int ns = ce - cb;
        const int m0 = mt * BM;
        const int n = nt * BN + wn0 + 4 * l31;               // the lane's pixel quad (tile rn holds pixel n + rn)
        const bool gok = n < p.Ntot;
        const int nn = gok ? n : 0;
        const int bi = nn / HW, pp = nn - bi * HW;
        const int64_t gcol = (int64_t)bi * p.YC * HW + pp;
        const bool partial = p.splits > 1;
        const int ks = __builtin_amdgcn_readfirstlane((item / p.n_mtiles) % p.splits);
        float* ybase = partial ? p.scratch + (int64_t)ks * p.y_elems : p.y;
        double* stt = (p.stats && !partial) ? p.stats + (int64_t)((nt * p.n_mtiles + mt) % p.nrep) * 2 * p.M : nullptr;
        const bool has_pre = PRE != 0 && !partial;
        float bias_pre[PRE ? WM : 1][PRE ? 16 : 1];        // the rows' bias (or 0), fetched now: the stage loop hides the round trip
        if (PRE) {
#pragma unroll
            for (int rm = 0; rm < (PRE ? WM : 0); ++rm)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    bias_pre[rm][reg] = (has_pre && p.bias) ? p.bias[gm < p.M ? gm : 0] : 0.0f;
                }
        }

        f32x16 acc[WM][4];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#pragma unroll 1
        for (int st = 0; st < ns; ++st, ++g) {
            const unsigned long long t0 = DMA_STAMP();
            wg_barrier();                                    // stage g is in LDS (the producers waited for their pieces)
            const unsigned long long t2 = DMA_STAMP();
            DMA_DBG_ADD(2, t2 - t0);
            DMA_DBG_ADD(7, 1);
            const float* sl = smem + __builtin_amdgcn_readfirstlane(g % NST) * ST_FL;
            const float* Aa = sl + a_lane;
            const float* Bb = sl + b_lane;
            const int kvalid = min(KCH, p.Ctot - (cb + st) * KCH);
            const int ng = (kvalid + 7) >> 3;                // k-groups of 8 channels = 4 k-steps (the mirror is zero past Ctot)
            // one loop over k-groups (see conv_dma.hip: two unrolled lengths made hipcc shuttle the accumulators through VGPRs).
            // Per group: WM reads of A (four k-steps each) and four reads of B (four tiles each), the B read of k-step s + 1 issued
            // among the MFMAs of k-step s; the A reads of the next group among the MFMAs of this group's last k-step.
            f32x4 a[WM], an[WM], b0, b1;
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = *reinterpret_cast<const f32x4*>(Aa + rm * 256);
            b0 = *reinterpret_cast<const f32x4*>(Bb);
#pragma unroll 1
            for (int kg = 0; kg < ng; ++kg) {
                const float* Ag = Aa + (kg + 1) * (BM * 8);  // next group (past the last one: read, never used)
                const float* Bg = Bb + kg * (8 * BN);
                auto mfmas = [&](int s, const f32x4& bb) {
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                        for (int rn = 0; rn < 4; ++rn) {
                            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm][s], bb[rn], acc[rm][rn], 0, 0, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                };
                __builtin_amdgcn_sched_barrier(0);
                b1 = *reinterpret_cast<const f32x4*>(Bg + 2 * BN);
                mfmas(0, b0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = *reinterpret_cast<const f32x4*>(Bg + 4 * BN);
                mfmas(1, b1);
                __builtin_amdgcn_sched_barrier(0);
                b1 = *reinterpret_cast<const f32x4*>(Bg + 6 * BN);
#pragma unroll
                for (int rm = 0; rm < WM; ++rm) an[rm] = *reinterpret_cast<const f32x4*>(Ag + rm * 256);
                mfmas(2, b0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = *reinterpret_cast<const f32x4*>(Bg + 8 * BN);      // k-step 0 of the next group
                mfmas(3, b1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rm = 0; rm < WM; ++rm) a[rm] = an[rm];
            }
            DMA_DBG_ADD(4, DMA_STAMP() - t2);
        }
        const unsigned long long t_epi = DMA_STAMP();

        // ---------------- epilogue: 16-byte pixel quads straight from the accumulators ---------------------------------------------------
        // residual / old value (accumulate): loaded here in batches of RB rows - all of a batch's loads are issued before the first is
        // used (one memory round trip per batch).  Prefetching them during the last stage (conv_dma.hip) would need 64 more registers
        // per 32 rows than two waves per SIMD leave; the bias (one scalar per row) IS prefetched, at the item's start.
        constexpr int RB = WM == 2 ? 4 : 8;
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int hb = 0; hb < 16 / RB; ++hb) {
                f32x4 add[PRE == 2 ? RB : 1];
                if (PRE == 2 && has_pre) {
                    // (each kind's loads in a loop of their own: a load next to its use makes hipcc wait for every element in turn)
                    f32x4 rv[RB], ov[RB];
                    int64_t off[RB];
#pragma unroll
                    for (int r8 = 0; r8 < RB; ++r8) {
                        const int reg = hb * RB + r8;
                        const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        const bool ok = gok && gm < p.M;
                        off[r8] = ok ? gcol + (int64_t)gm * HW : 0;      // an address that is always readable; the value is not used
                    }
                    if (p.res) {
#pragma unroll
                        for (int r8 = 0; r8 < RB; ++r8) rv[r8] = *reinterpret_cast<const f32x4*>(p.res + off[r8]);
                    }
                    if (p.beta) {
#pragma unroll
                        for (int r8 = 0; r8 < RB; ++r8) ov[r8] = *reinterpret_cast<const f32x4*>(p.y + off[r8]);
                    }
#pragma unroll
                    for (int r8 = 0; r8 < RB; ++r8) {
                        const float bsv = bias_pre[PRE ? rm : 0][PRE ? hb * RB + r8 : 0];
                        f32x4 v = {bsv, bsv, bsv, bsv};
                        if (p.res) { v[0] += rv[r8][0]; v[1] += rv[r8][1]; v[2] += rv[r8][2]; v[3] += rv[r8][3]; }
                        if (p.beta) { v[0] += ov[r8][0]; v[1] += ov[r8][1]; v[2] += ov[r8][2]; v[3] += ov[r8][3]; }
                        add[PRE == 2 ? r8 : 0] = v;
                    }
                }
#pragma unroll
                for (int r8 = 0; r8 < RB; ++r8) {
                    const int reg = hb * RB + r8;
                    const int gm = m0 + wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    const bool ok = gok && gm < p.M;
                    f32x4 v = {acc[rm][0][reg], acc[rm][1][reg], acc[rm][2][reg], acc[rm][3][reg]};
                    if (PRE == 2 && has_pre) { const f32x4 pv = add[PRE == 2 ? r8 : 0]; v[0] += pv[0]; v[1] += pv[1]; v[2] += pv[2]; v[3] += pv[3]; }
                    if (PRE == 1 && has_pre) { const float bsv = bias_pre[PRE ? rm : 0][PRE ? reg : 0]; v[0] += bsv; v[1] += bsv; v[2] += bsv; v[3] += bsv; }
                    float s = 0.0f, q = 0.0f;
                    if (ok) {
                        *reinterpret_cast<f32x4*>(ybase + gcol + (int64_t)gm * HW) = v;
                        s = (v[0] + v[1]) + (v[2] + v[3]);
                        q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
                    }
                    if (stt) {     // the row's sums over this wave's 128 pixels: the 32 lanes of a half hold them
                        s = half_sum_hi(s);
                        q = half_sum_hi(q);
                        if (l31 == 31 && gm < p.M) {
                            atomic_add_d(stt + gm, (double)s);
                            atomic_add_d(stt + p.M + gm, (double)q);
                        }
                    }
                }
            }
        DMA_DBG_ADD(5, DMA_STAMP() - t_epi);
    }
    DMA_DBG_ADD(0, DMA_STAMP() - t_begin);
    DMA_DBG_ADD(6, 1);
    DMA_DBG_FLUSH();
}

// -------------------------------------------------------------------------------------------------
template <int WM, int WVM, int WVN, int NST, int PRE, int KCH>
static int launch_q4(ConvP& p, int n_ntiles, int splits, hipStream_t st) {
    constexpr int BM = 32 * WM * WVM, BN = 128 * WVN;
    constexpr size_t lds = ((size_t)NST * KCH * (BM + BN) + 2 * BN + 64) * sizeof(float);   // (+ slack: the last k-group's look-ahead read of B, two rows past the ring)
    static_assert(lds <= 160 * 1024, "LDS image");
    p.n_mtiles = cdiv(p.M, BM);
    p.splits = splits;
    const int64_t items = (int64_t)p.n_mtiles * n_ntiles * splits;
    if (items <= 0 || items > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)items); return S2K_EINVAL; }
    p.n_tiles = (int)items;
    p.y_elems = (int64_t)p.B * p.YC * p.HO * p.WO;
    auto kern = conv_q4_kernel<WM, WVM, WVN, NST, PRE, KCH>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    static PerDeviceOnce cu_once;
    static int n_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    cu_once.run([&] {
        hipDeviceProp_t pr;
        n_cu[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    });
    static const int per_cu = tune_int("S2K_Q4_PER_CU", 1);
    const int grid = (int)std::min<int64_t>(items, (int64_t)n_cu[dev] * per_cu);     // one workgroup per CU (see conv_dma.hip)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, p);
    if (splits > 1) launch_splitk_reduce(p, st);
    g_s2k_variant = 4;
    return S2K_OK;
}

// S2K_OK = launched, 1 = not one of this kernel's shapes, < 0 = error
int launch_conv_q4(ConvP& p, hipStream_t st) {
    static const int enabled = tune_int("S2K_CONV_Q4", 1);
    if (!enabled || !p.wtq || p.res_mul) return 1;
    if (p.mode != S2K_MODE_CONV || p.KH != 1 || p.KW != 1 || p.S != 1 || p.C2 != 0 || p.gate1 || p.pro1 != S2K_PRO_NONE || p.x1_bf16) return 1;
    if ((p.HW & 3) || p.HO != p.H || p.WO != p.W) return 1;
    if (p.M < 24) return 1;
    // Where this kernel is used (rocprofv3 kernel durations, profiles/r04_q4_ab.md): the large maps' short reductions and thin layers
    // (>= 32 x 32: 240 x 40 49 vs 58 us on the generic kernel, 144 x 40 38 vs 44, 40 x 240 47 vs 54, 384 x 64 28 vs 29, 64 x 128 at
    // 128 x 128 117 vs 125) and the deepest 8 x 8 layers (512 x 3072: 70 vs 80 - 88).  On the 8 x 8 / 16 x 16 maps' other layers the
    // LDS-DMA ring kernel's 192- and 320-row tiles fit whole rounds of workgroups better (conv_dma.hip), the ViT Linears are at parity
    // with conv_pc_kernel (2304 x 768 over 3,328 tokens: 109 vs 112 us; 768 x 768: 57 vs 47) and stay there, and M < 40 pads a 64-row
    // wave tile too much.  S2K_FLAG_DMA beside S2K_FLAG_Q4 (tests) or S2K_CONV_Q4=2 (tuning builds): every supported shape.
    if (enabled != 2 && !p.force_dma &&
        !((p.Ntot >= 32768 && p.M >= 40 && p.Ctot < 256) || (p.Ntot <= 8192 && p.M <= 512 && p.Ctot >= 2048))) return 1;
    const int nchunks = cdiv(p.Ctot, 16);
    const int pre = (p.res || p.beta) ? 2 : (p.bias ? 1 : 0);
    // tile candidates (rows x pixels): 256 x 128, 128 x 256, 64 x 512 (wave tile 64 x 128), 128 x 128, 64 x 256 (wave tile 32 x 128)
    struct Cand { int wm, wvm, wvn; };
    static const Cand cands[5] = {{2, 4, 1}, {2, 2, 2}, {2, 1, 4}, {1, 4, 1}, {1, 2, 2}};
    static const int force = tune_int("S2K_Q4_TILE", -1), force_sp = tune_int("S2K_Q4_SPLITS", 0);
    int best = -1, best_sp = 1;
    double best_cost = 1e30;
    for (int ci = 0; ci < 5; ++ci) {
        if (force >= 0 && ci != force) continue;
        const Cand& c = cands[ci];
        if (pre == 2 && c.wm != 1) continue;                   // the epilogue's batched residual / old-value loads need 64 registers beside
                                                               // the accumulators: 32-row wave tiles only
        const int bm = 32 * c.wm * c.wvm, bn = 128 * c.wvn;
        const int nmt = cdiv(p.M, bm);
        if (nmt * bm > p.w_st) continue;                       // the mirror is zero padded to MP = w_st rows only
        const int64_t nnt = cdiv(p.Ntot, bn);
        const int64_t items = nmt * nnt;
        int sp = 1;
        if (p.scratch && items < 256 && nchunks >= 8) {
            sp = (int)std::min<int64_t>(8, std::min<int64_t>(nchunks / 4, cdiv64(256, items)));
            if (sp < 1) sp = 1;
        }
        if (force_sp > 0 && p.scratch) sp = std::min(force_sp, std::max(1, nchunks / 2));
        const double rounds = (double)cdiv64(items * sp, 256);
        const double per_item = (double)bm * bn * ((double)cdiv(nchunks, sp) + 2.0 + (sp > 1 ? 2.0 : 0.0));
        // the 32-row wave tile issues 5 reads per 16 MFMAs instead of 6 per 32: a little slower per MFMA
        const double cost = rounds * per_item * (c.wm == 1 ? 1.10 : 1.0);
        if (cost < best_cost) { best_cost = cost; best = ci; best_sp = sp; }
    }
    if (best < 0) return 1;
    const Cand& c = cands[best];
    const int bn = 128 * c.wvn;
    {   // 32-bit buffer offsets (descriptors are based at the first image a tile touches)
        const int64_t span = (p.HW % bn) == 0 ? 1 : std::min<int64_t>(p.B, (bn - 2) / p.HW + 2);
        const int64_t need = (int64_t)p.C1 * p.HW * 4 * span;
        if (need >= 0x7ffffff0ll) return 1;
    }
    const int nnt = cdiv(p.Ntot, bn);
    // stage depth: 32 channels in a ring of 3 where the LDS holds it (the per-stage barrier and the exposed latency of a stage's first
    // operand reads are paid half as often: 142 of ~2,200 cycles per 16-channel stage of a 128 x 128 tile), else 16 in a ring of 4
    static const int force_kch = tune_int("S2K_Q4_KCH", 0);
    const bool wide = c.wvn == 4;                              // 64 x 512 tile: 36 KB per 16-channel stage
    const bool k32 = !wide && (force_kch ? force_kch == 32 : p.Ctot > 32);
#define Q4_GO(WMv, WVMv, WVNv, PREv) (k32 ? launch_q4<WMv, WVMv, WVNv, 3, PREv, 32>(p, nnt, best_sp, st) : launch_q4<WMv, WVMv, WVNv, 4, PREv, 16>(p, nnt, best_sp, st))
#define Q4_CFG2(i, WMv, WVMv, WVNv) if (best == i) return pre == 1 ? Q4_GO(WMv, WVMv, WVNv, 1) : Q4_GO(WMv, WVMv, WVNv, 0);
#define Q4_CFG1(i, WMv, WVMv, WVNv) if (best == i) return pre == 2 ? Q4_GO(WMv, WVMv, WVNv, 2) : (pre == 1 ? Q4_GO(WMv, WVMv, WVNv, 1) : Q4_GO(WMv, WVMv, WVNv, 0));
    Q4_CFG2(0, 2, 4, 1) Q4_CFG2(1, 2, 2, 2) Q4_CFG1(3, 1, 4, 1) Q4_CFG1(4, 1, 2, 2)
    if (best == 2) return pre == 1 ? launch_q4<2, 1, 4, 4, 1, 16>(p, nnt, best_sp, st) : launch_q4<2, 1, 4, 4, 0, 16>(p, nnt, best_sp, st);
#undef Q4_CFG1
#undef Q4_CFG2
#undef Q4_GO
    return 1;
}

#if defined(S2K_TUNING) && defined(S2K_DMA_STAMPS)
extern "C" int s2k_debug_q4_counters(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_q4_dbg), sizeof(g_q4_dbg)) != hipSuccess) return S2K_EHIP;
    if (reset) {
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q4_dbg), z, sizeof(z));
    }
    return S2K_OK;
}
#endif

}  // namespace s2k
