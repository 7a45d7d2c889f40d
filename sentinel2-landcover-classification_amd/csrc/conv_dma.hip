// 1x1 convolution / Linear on the f32 matrix cores for inputs WITHOUT a load prologue, staged by LDS-DMA (round 4).
//
// Replaces, for the prologue-free 1x1 contractions of the hot path - the MBConv expand convs (reference
// efficientnet_unet.py:319-336, their input is the materialised block output), every 1x1 data gradient of the encoder
// (ATen convolution_backward of :319-372), the space-to-depth forms of the ConvTranspose backward, the ViT Linears'
// data gradients - what conv_igemm_kernel did at 32 % of the f32 MFMA peak.  Same arithmetic, operand layout (packed weights
// [k][MP] from WEIGHT_PACK, raw NCHW / feature-major activations), epilogue (bias, residual, accumulate, BatchNorm batch
// statistics, split-K partials for splitk_reduce_kernel) as igemm.hip.
//
// Why another kernel (measured, profiles/r04_*): these problems are 2 - 3 GFLOP each (15 - 20 us of matrix-core time) and the
// generic kernel spent 45 - 60 us on them: its K loop keeps ONE chunk in flight (global -> registers -> LDS), so a chunk costs
// max(MFMA time, memory latency) and with short chunks (KCH 16: 0.85 us of MFMAs against 1 - 2 us of latency) or short
// reductions (K = 24 .. 176: two or three chunks in all, nothing to pipeline) the matrix cores wait; its grid is one workgroup
// per tile, so every tile pays the pipeline fill and the epilogue, and 1.1 - 2.3 rounds of workgroups quantise badly.
//
// Structure:
//   * the operands go global -> LDS by `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write, one instruction per KiB): a ring of
//     NST stages of KCH = 16 channels {A = weights [k][BM], B = activations [k][BN]} with NST - 1 stages in flight - the
//     prefetch distance is 3 stages = thousands of cycles whatever the register budget;
//   * workgroups are PERSISTENT: a workgroup walks work items (m-tile, pixel tile, K split) and its stage stream runs across
//     item boundaries, so the first stages of the next item land while the current item's last stages and its epilogue run;
//   * one `s_barrier` per stage (raw barrier + counted `s_waitcnt vmcnt`: a `__syncthreads()` would drain the ring);
//   * the tile is BM = 64 * WM (WM 1..5) x BN = 64 * WN: M = 176 / 304 / 1056 run on 192 / 320 / 192-row tiles instead of padding
//     to a multiple of 128, and the launcher picks the tile by whole rounds of 256 workgroups;
//   * deep reductions over few pixels (8x8 / 16x16 maps) are cut along K into items of equal length (balanced partition) whose
//     partial tiles splitk_reduce_kernel adds in a fixed order (deterministic);
//   * the K tail is handled in half stages of 8 channels (K = 24 and 40 waste nothing).
#include <algorithm>

#include "common.h"
#define DMA_DBG_SYM g_dma_dbg
#include "dma.h"
#include "igemm.h"

namespace s2k {

// PRE: the stage has a bias, a residual or accumulates into Y (separate instantiation: the prefetched values double the
// accumulator registers, so only tiles of <= 5 accumulator tiles per wave carry it)
template <int WM, int WN, int NST, bool PRE, int KCH>
__global__ void __launch_bounds__(256) conv_dma_kernel(const ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_FL = KCH * BM, B_FL = KCH * BN, ST_FL = A_FL + B_FL;    // floats per stage
    constexpr int PA = A_FL / 1024, PB = B_FL / 1024, PW = PA + PB;        // 1-KiB pieces per wave per stage
    static_assert(KCH == 16 || KCH == 32, "stage depth");
    // transposed-store image (see igemm.hip), one PER WAVE: a wave's LDS operations execute in order, so a wave-private image
    // needs no workgroup barrier (the 64-row image shared by the workgroup cost two barriers per 64 rows: 4,800 cycles per
    // 192 x 64 item in all).  32 rows x (32 * WN + 4) floats; a lane reads a pixel quad, 64 / QW rows per instruction.
    constexpr int CW = 32 * WN + 4, CT_FL = 32 * CW;
    constexpr int QW = 8 * WN, RPI = 64 / QW, IT = 32 / RPI;
    static_assert(NST >= 3 && PW * (NST - 1) < 48, "ring depth vs the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ct = smem + NST * ST_FL + (threadIdx.x >> 6) * CT_FL;       // this wave's image

#ifdef S2K_TUNING
    const int exp_flags = p.exp;       // S2K_CV_EXP ablations: 1 = no epilogue, 2 = no MFMAs, 4 = no DMA, 8 = no stores, 16 = MFMA operands from registers (results are garbage)
#else
    constexpr int exp_flags = 0;
#endif
    DMA_DBG_DECL();
    const unsigned long long t_begin = DMA_STAMP();
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * (WM * 32), wn0 = (wave & 1) * (WN * 32);
    const int HW = p.HW;
    const int nchunks = (p.Ctot + KCH - 1) / KCH;
    const int n_items = p.n_tiles;                       // n_ntiles * splits * n_mtiles; item = (nt * splits + ks) * n_mtiles + mt
    const int pos = xcd_remap(blockIdx.x, gridDim.x);    // consecutive items (the m-tiles / K splits of one pixel tile) share an XCD's L2

    // ---- this wave's pieces: piece q = wave + 4 j covers stage floats [256 q, 256 q + 256); a lane moves 4 of them --------------
    uint32_t a_fix[PA];          // byte offset of the lane's quad inside the A tile: (row * w_st + col) * 4
    int b_row[PB], b_col[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int f = 256 * (wave + 4 * j) + 4 * lane;
        a_fix[j] = (uint32_t)((f / BM) * p.w_st + (f % BM)) * 4u;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int f = 256 * (wave + 4 * j) + 4 * lane;
        b_row[j] = f / BN;
        b_col[j] = f % BN;
    }
    const rsrc_t rw = make_rsrc(p.wt, 0x7ffffff0ll);     // the WPACK entry is zero padded to [KP][MP]: no range check needed
    const uint32_t hw4 = (uint32_t)HW * 4u;

    // ---- producer state (runs up to NST - 1 stages ahead of the consumer, possibly in the next item) ---------------------------
    int p_item = pos, p_st = 0, p_ns = 0, p_c0 = 0;
    uint32_t p_avoff[PA], p_bpix[PB];
    rsrc_t p_rx = make_rsrc(p.x1, 0);
    auto item_range = [&](int item, int& mt, int& nt, int& cb, int& ce) {
        // (integer division runs on the vector ALU: tell the compiler that the results are wave-uniform, or every DMA below gets a
        // waterfall loop around its scalar offset)
        const int r = item / p.n_mtiles;
        mt = __builtin_amdgcn_readfirstlane(item - r * p.n_mtiles);
        nt = __builtin_amdgcn_readfirstlane(r / p.splits);
        const int ks = r - nt * p.splits;
        cb = __builtin_amdgcn_readfirstlane((ks * nchunks) / p.splits);
        ce = __builtin_amdgcn_readfirstlane(((ks + 1) * nchunks) / p.splits);
    };
    auto producer_enter = [&]() {        // geometry of item p_item (wave-uniform control flow)
        int mt, nt, cb, ce;
        item_range(p_item, mt, nt, cb, ce);
        p_ns = ce - cb;
        p_c0 = cb * KCH;
        p_st = 0;
        const int n0 = nt * BN;
        const int img_b = __builtin_amdgcn_readfirstlane(n0 / HW);       // descriptor based at the tile's first image (32-bit offsets, see igemm.hip)
        p_rx = make_rsrc(p.x1 + (int64_t)img_b * p.C1 * HW, (int64_t)(p.B - img_b) * p.C1 * HW * 4);
#pragma unroll
        for (int j = 0; j < PA; ++j) p_avoff[j] = a_fix[j] + (uint32_t)(mt * BM) * 4u;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int n = n0 + b_col[j];
            const int b = n / HW, pp = n - b * HW;
            p_bpix[j] = n < p.Ntot ? (uint32_t)((int64_t)(b - img_b) * p.C1 * HW + pp) * 4u : BUF_OOB;
        }
    };
    int issued = 0;                      // stages issued so far (= global stream index of the next stage to issue)
    bool p_done = false;                 // this workgroup's stage stream is exhausted
    auto issue = [&]() {                 // invariant while !p_done: p_item is a valid item and p_st < p_ns
        float* slot = smem + __builtin_amdgcn_readfirstlane(issued % NST) * ST_FL;
        const int c0 = __builtin_amdgcn_readfirstlane(p_c0 + p_st * KCH);
        const uint32_t soa = (uint32_t)c0 * (uint32_t)p.w_st * 4u;
        const uint32_t sob = (uint32_t)c0 * hw4;
        const int cmax = p.Ctot - 1 - c0;                    // rows past the last channel re-read it (they meet zero weight rows)
        if (!(exp_flags & 4)) {
#pragma unroll
            for (int j = 0; j < PA; ++j)
                dma16(rw, slot + 256 * (wave + 4 * j), p_avoff[j], soa);
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const uint32_t voff = p_bpix[j] + (uint32_t)min(b_row[j], cmax) * hw4;    // out of range + row offset stays out of range
                dma16(p_rx, slot + A_FL + 256 * (wave + 4 * j), voff, sob);
            }
        }
        ++issued;
        if (++p_st == p_ns) {
            p_item += gridDim.x;
            if (p_item < n_items) producer_enter();
            else p_done = true;
        }
    };
    if (p_item >= n_items) return;       // (the launcher never starts more workgroups than items)
    producer_enter();
#pragma unroll 1
    for (int i = 0; i < NST - 1 && !p_done; ++i) issue();

    DMA_DBG_ADD(1, DMA_STAMP() - t_begin);
    // ---- consumer -----------------------------------------------------------------------------------------------------------------
    const int a_lane = lh * BM + wm0 + l31;                  // + (2 s) * BM + rm * 32
    const int b_lane = A_FL + lh * BN + wn0 + l31;           // + (2 s) * BN + rn * 32
    int g = 0;                                               // global index of the stage being consumed
    int skip = 0;                                            // stages from g on whose pieces are known to have landed
#pragma unroll 1
    for (int item = pos; item < n_items; item += gridDim.x) {
        int mt, nt, cb, ce;
        item_range(item, mt, nt, cb, ce);
        const int ns = ce - cb;
        // ---- geometry of the transposed store: lane = pixel quad q4 of rows r0, r0 + RPI, ... of each of the wave's 32-row tiles --
        const int m0 = mt * BM;
        const int q4 = lane % QW, r0 = lane / QW;
        const int n = nt * BN + wn0 + 4 * q4;
        const bool gok = n < p.Ntot;
        const int nn = gok ? n : 0;
        const int bi = nn / HW, pp = nn - bi * HW;
        const int64_t gcol = (int64_t)bi * p.YC * HW + pp;
        const bool partial = p.splits > 1;
        const int ks = __builtin_amdgcn_readfirstlane((item / p.n_mtiles) % p.splits);
        float* ybase = partial ? p.scratch + (int64_t)ks * p.y_elems : p.y;
        double* stt = (p.stats && !partial) ? p.stats + (int64_t)((nt * p.n_mtiles + mt) % p.nrep) * 2 * p.M : nullptr;
        // bias + residual + old value (accumulate), fetched to registers during the LAST stage of the item: an ordinary load beside
        // LDS-DMAs makes hipcc wait for vmcnt(0) at its first use - issued here, that wait falls behind a stage of MFMAs and every
        // DMA it drains is older than the loads anyway
        const bool has_pre = PRE && !partial;
        f32x4 pre[PRE ? WM : 1][PRE ? IT : 1];

        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#pragma unroll 1
        for (int st = 0; st < ns; ++st, ++g) {
            // my pieces of stage g have landed once at most (issued - g - 1) younger stages are outstanding.  vmcnt counts the
            // epilogue's stores and atomics too, in issue order: a counted wait with such operations YOUNGER than the awaited pieces
            // would have to know how many of them were issued (exec-masked stores may be skipped), and a count that is too small makes
            // the wave wait for its stores to retire (measured: 14 us per 256 x 128 item instead of 6).  So the ring is drained up to
            // its youngest stage BEFORE every epilogue (below) and the stages already known to have landed are not waited for again.
            const unsigned long long t0 = DMA_STAMP();
            if (skip > 0) --skip;
            else if (issued - g - 1 >= NST - 2) wait_vm<PW * (NST - 2)>();
            else wait_vm<0>();
            wg_barrier();                                    // everyone's pieces of stage g are in LDS; slot (g - 1) % NST is free
            const unsigned long long t1 = DMA_STAMP();
            if (!p_done) issue();
            const unsigned long long t2 = DMA_STAMP();
            DMA_DBG_ADD(2, t1 - t0);
            DMA_DBG_ADD(3, t2 - t1);
            DMA_DBG_ADD(7, 1);
            if (PRE && has_pre && st == ns - 1) {
#pragma unroll
                for (int ps = 0; ps < (PRE ? WM : 0); ++ps)
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int r = r0 + it * RPI;
                        const int gm = m0 + wm0 + ps * 32 + r;
                        const bool ok = gok && gm < p.M;
                        const int64_t off = ok ? gcol + (int64_t)gm * HW : 0;      // an address that is always readable; the value is not used
                        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (p.bias) { const float bsv = p.bias[ok ? gm : 0]; v[0] = bsv; v[1] = bsv; v[2] = bsv; v[3] = bsv; }
                        if (p.res) { const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + off); v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3]; }
                        if (p.beta) { const f32x4 ov = *reinterpret_cast<const f32x4*>(p.y + off); v[0] += ov[0]; v[1] += ov[1]; v[2] += ov[2]; v[3] += ov[3]; }
                        pre[ps][it] = v;
                    }
            }
            const float* sl = smem + __builtin_amdgcn_readfirstlane(g % NST) * ST_FL;
            const float* Aa = sl + a_lane;
            const float* Bb = sl + b_lane;
            const int kvalid = min(KCH, p.Ctot - (cb + st) * KCH);    // K tail: half stages of 8 channels
            // Explicit two-set operand pipeline (as conv_pc_kernel's consumers): the LDS reads of k-step s + 1 are issued between the
            // MFMAs of k-step s.  ONE loop over half stages of 4 k-steps (a full stage = 2 trips, the K tail = 1): with the two
            // lengths as two unrolled branches hipcc kept the accumulators in VGPRs across the stage loop and copied all of them
            // into AGPRs and back around every stage (96 v_accvgpr moves per 24 MFMAs on the 192 x 64 tile).
            if (exp_flags & 2) continue;
            {
                auto lds_ops = [&](const float* Ah, const float* Bh, int s, float (&a)[WM], float (&b)[WN]) {
                    if (exp_flags & 16) {      // ablation: operands from registers (what the bare MFMA stream of this loop costs)
#pragma unroll
                        for (int rm = 0; rm < WM; ++rm) a[rm] = __builtin_bit_cast(float, (int)(s + rm + lane));
#pragma unroll
                        for (int rn = 0; rn < WN; ++rn) b[rn] = __builtin_bit_cast(float, (int)(s - rn + lane));
                        return;
                    }
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm) a[rm] = Ah[(2 * s) * BM + rm * 32];
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn) b[rn] = Bh[(2 * s) * BN + rn * 32];
                };
                auto mfmas = [&](const float (&a)[WM], const float (&b)[WN]) {
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm)
#pragma unroll
                        for (int rn = 0; rn < WN; ++rn)
                            acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm], b[rn], acc[rm][rn], 0, 0, 0);
                };
                auto interleave = [&]() {
#pragma unroll
                    for (int t = 0; t < WM * WN; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // <= 2 DS reads
                    }
                };
                const int nh = (kvalid + 7) >> 3;          // half stages of 8 channels = 4 k-steps
                float a0[WM], a1[WM], b0[WN], b1[WN];
                lds_ops(Aa, Bb, 0, a0, b0);
#pragma unroll 1
                for (int h = 0; h < nh; ++h) {
                    const float* Ah = Aa + h * (8 * BM);
                    const float* Bh = Bb + h * (8 * BN);
#pragma unroll
                    for (int s = 0; s < 4; s += 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        lds_ops(Ah, Bh, s + 1, a1, b1);
                        mfmas(a0, b0);
                        interleave();
                        __builtin_amdgcn_sched_barrier(0);
                        lds_ops(Ah, Bh, s + 2, a0, b0);     // s + 2 = 4: k-step 0 of the next half (past the last one: read, never used)
                        mfmas(a1, b1);
                        interleave();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            DMA_DBG_ADD(4, DMA_STAMP() - t2);
        }

        // ---------------- epilogue: transposed store through the ct image (igemm.hip), raw barriers only -------------------------
        if (exp_flags & 1) continue;
        const unsigned long long t_epi = DMA_STAMP();
        {   // every stage but the youngest has landed before the first store is issued (see the wait above)
            const int ahead = issued - g;
            if (ahead >= 1) { wait_vm<PW>(); skip = ahead - 1; }
            else skip = 0;
        }
#pragma unroll
        for (int ps = 0; ps < WM; ++ps) {                    // the wave's accumulator row ps: 32 rows x 32 * WN pixels
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) ct[rl * CW + rn * 32 + l31] = acc[ps][rn][reg];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = r0 + it * RPI;
                const int gm = m0 + wm0 + ps * 32 + r;
                const bool ok = gok && gm < p.M;
                f32x4 v = *reinterpret_cast<const f32x4*>(ct + r * CW + 4 * q4);
                float s = 0.0f, q = 0.0f;
                if (PRE && has_pre) { const f32x4 pv = pre[PRE ? ps : 0][PRE ? it : 0]; v[0] += pv[0]; v[1] += pv[1]; v[2] += pv[2]; v[3] += pv[3]; }
                if (ok) {
                    if (!(exp_flags & 8)) *reinterpret_cast<f32x4*>(ybase + gcol + (int64_t)gm * HW) = v;
                    s = (v[0] + v[1]) + (v[2] + v[3]);
                    q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
                }
                if (stt) {     // a row's sums over this wave's pixels: one DPP reduction over the QW lanes that hold them, one f64 atomic
                    bool writer;       // pair per row, wave column and item
                    if (QW == 8) {
                        s += dpp_mov0<DPP_XOR1>(s); s += dpp_mov0<DPP_XOR2>(s); s += dpp_mov0<DPP_HALF_MIRROR>(s);
                        q += dpp_mov0<DPP_XOR1>(q); q += dpp_mov0<DPP_XOR2>(q); q += dpp_mov0<DPP_HALF_MIRROR>(q);
                        writer = (lane & 7) == 7;
                    } else { s = row16_sum(s); q = row16_sum(q); writer = (lane & 15) == 15; }
                    if (writer && gm < p.M) {
                        atomic_add_d(stt + gm, (double)s);
                        atomic_add_d(stt + p.M + gm, (double)q);
                    }
                }
            }
        }
        DMA_DBG_ADD(5, DMA_STAMP() - t_epi);
    }
    wait_vm<0>();
    DMA_DBG_ADD(0, DMA_STAMP() - t_begin);
    DMA_DBG_ADD(6, 1);
    DMA_DBG_FLUSH();
}

// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int NST, bool PRE, int KCH>
static int launch_dma(ConvP& p, int n_ntiles, int splits, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr size_t lds = ((size_t)NST * KCH * (BM + BN) + (size_t)4 * 32 * (32 * WN + 4)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS image");
    p.n_mtiles = cdiv(p.M, BM);
    p.splits = splits;
    const int64_t items = (int64_t)p.n_mtiles * n_ntiles * splits;
    if (items <= 0 || items > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)items); return S2K_EINVAL; }
    p.n_tiles = (int)items;
    p.y_elems = (int64_t)p.B * p.YC * p.HO * p.WO;
    auto kern = conv_dma_kernel<WM, WN, NST, PRE, KCH>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // persistent grid = the workgroups that are resident at once (registers and LDS decide: 1 or 2 per CU)
    static PerDeviceOnce cu_once;
    static int n_slots[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    cu_once.run([&] {
        hipDeviceProp_t pr;
        const int cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        int per_cu = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        // ONE workgroup per CU: a wave that shares its SIMD with a streaming f32-MFMA wave gets about one instruction issued per
        // MFMA (wgrad_pc.hip), so a second workgroup's epilogue and DMA issue take 10x longer than alone (measured: 40 k cycles per
        // epilogue instead of 4.8 k) and its barriers hold its MFMA waves back - two per CU ran 8 % slower than one
        static const int want = tune_int("S2K_DMA_PER_CU", 1);
        n_slots[dev] = cus * std::max(1, std::min(per_cu, want));
    });
    const int grid = (int)std::min<int64_t>(items, n_slots[dev]);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
    if (splits > 1) launch_splitk_reduce(p, st);
    g_s2k_variant = 3;
    return S2K_OK;
}

// S2K_OK = launched, 1 = not one of this kernel's shapes, < 0 = error
int launch_conv_dma(ConvP& p, hipStream_t st) {
    static const int enabled = tune_int("S2K_CONV_DMA", 1);
    if (!enabled || p.res_mul) return 1;
    if (p.mode != S2K_MODE_CONV || p.KH != 1 || p.KW != 1 || p.S != 1 || p.C2 != 0 || p.gate1 || p.pro1 != S2K_PRO_NONE || p.x1_bf16) return 1;
    if ((p.HW & 3) || p.HO != p.H || p.WO != p.W) return 1;
    if (p.M < 40) return 1;                                   // thin layers at full resolution: the wide-pixel tiles of igemm.hip
    // Where this kernel is used (rocprofv3 kernel durations against the kernels it replaces, profiles/r04_dma_ab.md): the 8x8 and
    // 16x16 maps' deep or wide layers - 128 x 768 (25 vs 34 us with the split-K tail), 768 x 176 (32 vs 37), 1056 x 176 (49 vs 52),
    // 768 x 128 (26 vs 28), 304 x 1824 (41 vs 44), 512 x 2048 (59 vs 63).  On the larger maps the short reductions (240 x 40 at
    // 64x64, 384 x 64 at 32x32, 40 x 240) tie or lose by 5 - 20 %: they are bound by the serial epilogue of a write-heavy tile,
    // which one workgroup per CU cannot hide, so they stay on the generic kernel (two workgroups per CU).  S2K_CONV_DMA=2: everything.
    if (enabled != 2 && !p.force_dma && !(p.Ntot <= 8192 && (p.Ctot >= 512 || p.M >= 768))) return 1;
    const int nchunks = cdiv(p.Ctot, 16);
    // tile: BM = 64 * WM rows x BN = 64 * WN pixels.  Cost of a candidate = rounds of 256 workgroups x tile area (the matrix-core
    // time of the slowest CU), padding included; K is cut (<= 8 ways, >= 4 chunks per cut) only when the items do not fill the chip.
    static const int force_wm = tune_int("S2K_DMA_WM", 0), force_wn = tune_int("S2K_DMA_WN", 0), force_sp = tune_int("S2K_DMA_SPLITS", 0);
    const bool pre = p.bias || p.res || p.beta;
    int best_wm = 0, best_wn = 0, best_sp = 1;
    double best_cost = 1e30;
    for (int wm = 1; wm <= 5; ++wm) {
        const int bm = 64 * wm, nmt = cdiv(p.M, bm);
        if (nmt * bm > p.w_st) continue;                       // the packed weights are zero padded to MP = w_st columns only
        for (int wn = 1; wn <= 2; ++wn) {
            if (wm == 5 && wn == 2) continue;                  // 160 accumulator registers: not instantiated
            if (pre && wm * wn > 5) continue;                  // PRE variants: see the kernel
            if (force_wm && (wm != force_wm || (force_wn && wn != force_wn))) continue;
            const int64_t nnt = cdiv(p.Ntot, 64 * wn);
            const int64_t items = nmt * nnt;
            int sp = 1;
            if (p.scratch && items < 256 && nchunks >= 8) {
                sp = (int)std::min<int64_t>(8, std::min<int64_t>(nchunks / 4, cdiv64(256, items)));
                if (sp < 1) sp = 1;
            }
            if (force_sp > 0 && p.scratch) sp = std::min(force_sp, std::max(1, nchunks / 2));
            const double rounds = (double)cdiv64(items * sp, 256);
            // per-item time ~ tile area x chunks (+ a fixed cost per item: epilogue / fill, in units of chunks)
            const double per_item = (double)bm * 64 * wn * ((double)cdiv(nchunks, sp) + 3.0 + (sp > 1 ? 2.0 : 0.0));
            // smaller tiles re-read the other operand more often: a mild preference for the larger tile at equal cost
            const double cost = rounds * per_item * (1.0 + 0.02 * (6 - wm) + 0.02 * (2 - wn));
            if (cost < best_cost) { best_cost = cost; best_wm = wm; best_wn = wn; best_sp = sp; }
        }
    }
    if (!best_wm) return 1;
    {   // 32-bit buffer offsets (descriptors are based at the first image a tile touches)
        const int bn = 64 * best_wn;
        const int64_t span = (p.HW % bn) == 0 ? 1 : std::min<int64_t>(p.B, (bn - 2) / p.HW + 2);
        const int64_t need = (int64_t)p.C1 * p.HW * 4 * span;
        if (need >= 0x7ffffff0ll) return 1;                    // the generic path reports what cannot be addressed
    }
    const int nnt = cdiv(p.Ntot, 64 * best_wn);
    // stage depth: 32 channels (ring of 3) wherever the LDS holds it - the per-stage costs (barrier, counted wait, the exposed LDS
    // latency of a stage's first operands, loop control: ~350 cycles) are paid half as often; the two largest tiles keep 16-channel
    // stages in a ring of 4
    static const int force_kch = tune_int("S2K_DMA_KCH", 0);
    const bool big = best_wm == 5 || (best_wm == 4 && best_wn == 2);
    const bool k32 = !big && (force_kch ? force_kch == 32 : p.Ctot > 32);
#define DMA_GO(WMv, WNv, PREv) (k32 ? launch_dma<WMv, WNv, 3, PREv, 32>(p, nnt, best_sp, st) : launch_dma<WMv, WNv, 4, PREv, 16>(p, nnt, best_sp, st))
#define DMA_BIG(WMv, WNv, PREv) launch_dma<WMv, WNv, 4, PREv, 16>(p, nnt, best_sp, st)
#define DMA_CFG(WMv, WNv) if (best_wm == WMv && best_wn == WNv) return pre ? DMA_GO(WMv, WNv, true) : DMA_GO(WMv, WNv, false);
#define DMA_CFG_NOPRE(WMv, WNv) if (best_wm == WMv && best_wn == WNv && !pre) return DMA_GO(WMv, WNv, false);
    DMA_CFG(1, 1) DMA_CFG(1, 2)
    DMA_CFG(2, 1) DMA_CFG(2, 2)
    DMA_CFG(3, 1) DMA_CFG_NOPRE(3, 2)
    DMA_CFG(4, 1)
    if (best_wm == 4 && best_wn == 2 && !pre) return DMA_BIG(4, 2, false);
    if (best_wm == 5 && best_wn == 1) return pre ? DMA_BIG(5, 1, true) : DMA_BIG(5, 1, false);
#undef DMA_CFG_NOPRE
#undef DMA_GO
#undef DMA_BIG
#undef DMA_CFG
    return 1;
}

#if defined(S2K_TUNING) && defined(S2K_DMA_STAMPS)
extern "C" int s2k_debug_dma_counters(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dma_dbg), sizeof(g_dma_dbg)) != hipSuccess) return S2K_EHIP;
    if (reset) {
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dma_dbg), z, sizeof(z));
    }
    return S2K_OK;
}
#endif

}  // namespace s2k
