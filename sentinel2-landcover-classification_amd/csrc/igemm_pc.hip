// Implicit-GEMM convolution, producer / consumer form: the prologue-light, MFMA-bound dense convs of the hot path
// (1x1 convs / Linears and 3x3 stride-1 convs whose input carries no prologue or BatchNorm + ReLU: every data gradient,
// the decoder's double convs, the MBConv expand convs, every Linear of the ViT blocks).  Same arithmetic, operand layout
// (packed weights from WEIGHT_PACK, raw NCHW activations) and epilogue (bias, residual, accumulate, BatchNorm batch
// statistics) as igemm.hip, which keeps the SiLU / SE-gated, strided, transposed and split-K shapes; replaces the same
// reference code (efficientnet_unet.py:168-176,288-297,319-372, timm Block Linears of prithvi.py:162-183).
//
// Who does what (see wgrad_pc.hip for the measurements behind it):
//   * waves 0-3: CONSUMERS, 2 (m) x 2 (n), each WM x 2 accumulator tiles of 32 x 32: LDS reads + MFMAs only, fully unrolled
//     over the K chunk with every LDS offset an instruction immediate (tile geometry is a template parameter);
//   * waves 4-7: PRODUCERS: the next K chunk global -> registers -> (BatchNorm + ReLU) -> the other half of a double-buffered
//     LDS image {A = weights [k][m], B = activations [k][pixels] or [channel][halo tile]}.  A wave beside a back-to-back f32
//     MFMA stream gets about one instruction issued per MFMA, so the producers move 16 bytes per instruction wherever the
//     layout allows (weights always; activations of 1x1 convs: four pixels per lane) and do two vector instructions per
//     element at most;
//   * one barrier per K chunk: producer P(0) | P(1) | ...; consumer | C(0) | C(1) | ...
#include <algorithm>

#include "common.h"
#include "igemm.h"

namespace s2k {

// THIN (3x3 only): M <= 32 - one 32-row m-tile shared by the four consumer waves, which split a 256-pixel tile 1 (m) x 4 (n)
// (the decoder's full-resolution 32-channel convs and their data gradients: a quarter of a 128-row tile would be padding)
// NARROW (1x1 only): 64 x 64 tiles (each consumer wave one 32 x 32 tile) for problems whose 128-pixel tiles would leave CUs
// empty - the ViT encoder's Linears over 3,328 visible tokens are 156 tiles of 128 x 128 on 256 CUs
template <int BMODE, int WM, int KCH, int R, int XW, int PRO, bool THIN = false, bool NARROW = false>
__global__ void __launch_bounds__(512) conv_pc_kernel(const ConvP p) {
    constexpr int NT = 256;
    constexpr int TT = (BMODE == BM_PIX) ? 1 : 9;
    constexpr int WN = NARROW ? 1 : 2;
    constexpr int BM = THIN ? 32 : WM * 64, BN = THIN ? 256 : (NARROW ? 64 : 128);
    constexpr int RPPB = NT / (BN / 4);                     // PIX: channel rows of B per producer pass
    static_assert(!NARROW || (BMODE == BM_PIX && WM == 1 && !THIN), "narrow tiles: 1x1, 64 rows");
    static_assert(BMODE != BM_PIX || KCH % RPPB == 0, "chunk rows per pass");
    constexpr int NWCOL = THIN ? 4 : 2;                    // wave columns (n) of the consumer grid
    static_assert(!THIN || (WM == 1 && BMODE == BM_SPATIAL), "thin tiles: 3x3, one m-tile of 32 rows");
    constexpr int KT = KCH * TT;                           // rows of an A chunk
    constexpr int WS = XW + 2, IR = R + 2;                 // 3x3 stride-1 halo tile
    constexpr int USED = IR * WS;
    constexpr int CSB = (BMODE == BM_PIX) ? BN : USED;     // B row stride (floats)
    constexpr int EPT = (USED + NT - 1) / NT;              // halo elements per producer thread and channel
    constexpr int A_FLOATS = KT * BM;
    constexpr int B_FLOATS = KCH * CSB;
    constexpr int BUF = (A_FLOATS + B_FLOATS + 3) & ~3;
    constexpr int A_RPP = NT / (BM / 4);                   // A rows per producer pass (a lane moves one float4)
    constexpr int NA = (KT + A_RPP - 1) / A_RPP;
    constexpr int KS = KCH / 2;                            // k-steps (channel pairs) per tap
    static_assert(BMODE == BM_PIX || R * XW <= BN, "3x3 tile");
    static_assert(KCH % 2 == 0 && (BMODE != BM_PIX || KCH % 8 == 0), "chunk shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool producer = threadIdx.x >= NT;
    const int tid = threadIdx.x & (NT - 1);
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int m0 = mt * BM;
    const int HWo = p.HO * p.WO;
    const int nchunks = (p.Ctot + KCH - 1) / KCH;
    // 3x3: tile origin
    const int tx = (BMODE == BM_PIX) ? 0 : nt % p.tiles_x;
    const int ty = (BMODE == BM_PIX) ? 0 : (nt / p.tiles_x) % p.tiles_y;
    const int sb = (BMODE == BM_PIX) ? 0 : nt / (p.tiles_x * p.tiles_y);
    const int y0 = ty * R, x0 = tx * XW;

    // ---- transposed store (both roles; see conv_bf16.hip): the consumers write their accumulator tiles to LDS 32 rows per wave row
    // at a time, then all 512 threads read them back pixel-major and store 16-byte pixel quads (an instruction = rows x 512 B instead
    // of 2 rows x 128 B straight from the MFMA layout: the write-heavy layers streamed their output at ~3 TB/s where a fill reaches
    // 6.8).  Bias / residual / accumulate and the BatchNorm statistics are applied on the reading side.
    constexpr bool TEPI = BMODE == BM_PIX || XW % 4 == 0;
    constexpr int CTP = BN + 8;                                  // LDS row stride: lane halves (rows r, r + 4) land in different banks
    constexpr int PASS_ROWS = THIN ? 32 : 64;
    constexpr int G = BN / 4, RPI = 2 * NT / G;
    static_assert(!TEPI || (PASS_ROWS % RPI == 0 && (G == 16 || G == 32 || G == 64)), "transposed epilogue geometry");   // (launch_pc sizes the LDS for it)
    auto store_rows = [&](int rm) {
        const int t = threadIdx.x, g4 = t % G, r0 = t / G;
        const int ln = t & 63;
        bool gok;
        int64_t gcol;
        const int j = 4 * g4;
        if (BMODE == BM_PIX) {
            const int n = nt * BN + j;
            gok = n < p.Ntot;
            const int nn = gok ? n : 0;
            const int b = nn / p.HW, pp = nn - b * p.HW;
            gcol = (int64_t)b * p.YC * HWo + pp;
        } else {
            const int r = j / XW, xx = j % XW;
            gok = (j < R * XW) && (y0 + r < p.HO) && (x0 + xx < p.WO);
            gcol = (int64_t)sb * p.YC * HWo + (int64_t)(y0 + r) * p.WO + (x0 + xx);
        }
        double* st = p.stats ? p.stats + (int64_t)(tile % p.nrep) * 2 * p.M : nullptr;
#pragma unroll
        for (int r = r0; r < PASS_ROWS; r += RPI) {
            const int gm = m0 + (r >> 5) * (WM * 32) + rm * 32 + (r & 31);
            const bool ok = gok && gm < p.M;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * CTP + 4 * g4);
            float s = 0.0f, q = 0.0f;
            if (ok) {
                if (p.bias) { const float bsv = p.bias[gm]; v[0] += bsv; v[1] += bsv; v[2] += bsv; v[3] += bsv; }
                float* dst = p.y + gcol + (int64_t)gm * HWo;
                if (p.res) { const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + gcol + (int64_t)gm * HWo); v[0] = res_combine(v[0], rv[0], p.res_mul); v[1] = res_combine(v[1], rv[1], p.res_mul); v[2] = res_combine(v[2], rv[2], p.res_mul); v[3] = res_combine(v[3], rv[3], p.res_mul); }
                if (p.beta) { const f32x4 ov = *reinterpret_cast<const f32x4*>(dst); v[0] += ov[0]; v[1] += ov[1]; v[2] += ov[2]; v[3] += ov[3]; }
                *reinterpret_cast<f32x4*>(dst) = v;
                s = (v[0] + v[1]) + (v[2] + v[3]);
                q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
            }
            if (st) {      // a row's sums: one DPP reduction over the lanes that hold it, one f64 atomic pair per row and workgroup
                bool writer;
                if (G == 16) { s = row16_sum(s); q = row16_sum(q); writer = (ln & 15) == 15; }
                else if (G == 32) { s = half_sum_hi(s); q = half_sum_hi(q); writer = (ln & 31) == 31; }
                else { s = wave_sum_hi(s); q = wave_sum_hi(q); writer = ln == 63; }
                if (writer && gm < p.M) {
                    atomic_add_d(st + gm, (double)s);
                    atomic_add_d(st + p.M + gm, (double)q);
                }
            }
        }
    };

    if (producer) {
        // =================================================================================================================
        // PRODUCER
        // =================================================================================================================
        const float* a_src = p.wt + (int64_t)(tid / (BM / 4)) * p.w_st + m0 + 4 * (tid % (BM / 4));
        f32x4 areg[NA];
        f32x4 bvec[(BMODE == BM_PIX) ? KCH / RPPB : 1];
        float breg[(BMODE == BM_PIX) ? 1 : KCH * EPT];
        // ---- geometry of this workgroup's B tile: fixed for the whole K loop ----
        uint32_t bvoff = BUF_OOB;            // PIX: byte offset of this lane's pixel quad (image-relative) or out of range
        uint32_t goff[EPT];                  // SPATIAL: byte offset of the halo element inside a channel plane, or out of range
        float bound[EPT];                    // +inf inside the image, 0 in the zero padding (ReLU + mask = one v_med3)
        int img_b = 0;
        const int j4 = tid % (BN / 4), kc0 = tid / (BN / 4);      // PIX: pixel quad, first channel row
        if (BMODE == BM_PIX) {
            const int n = nt * BN + 4 * j4;
            const bool ok = n < p.Ntot;
            img_b = (nt * BN) / p.HW;         // descriptors based at the tile's first image (see igemm.hip)
            const int nn = ok ? n : 0;
            const int b = nn / p.HW, pp = nn - b * p.HW;
            bvoff = ok ? (uint32_t)((int64_t)(b - img_b) * p.C1 * p.HW + pp) * 4u : BUF_OOB;
        } else {
            img_b = sb;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int e = tid + NT * i;
                const int iy = y0 - p.PT + e / WS, ix = x0 - p.PL + e % WS;
                const bool ok = e < USED && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                goff[i] = ok ? (uint32_t)(iy * p.W + ix) * 4u : BUF_OOB;
                bound[i] = ok ? __builtin_inff() : 0.0f;
            }
        }
        const int64_t x1_img = (int64_t)p.C1 * p.HW, x2_img = (int64_t)p.C2 * p.HW;
        const rsrc_t rx1 = make_rsrc(p.x1 + img_b * x1_img, (p.B - img_b) * x1_img * 4);
        const rsrc_t rx2 = make_rsrc(p.x2 ? p.x2 + img_b * x2_img : p.x1, p.x2 ? (p.B - img_b) * x2_img * 4 : 0);
        const uint32_t cs4 = (uint32_t)p.HW * 4u;

        auto fetch = [&](int c0) {
            const float* src = a_src + (int64_t)c0 * TT * p.w_st;
#pragma unroll
            for (int i = 0; i < NA; ++i)     // rows past the chunk (last partial pass) are read, never stored: the WPACK buffer has slack
                areg[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)i * A_RPP * p.w_st);
            if (BMODE == BM_PIX) {
#pragma unroll
                for (int i = 0; i < KCH / RPPB; ++i) {
                    const int c = min(c0 + kc0 + RPPB * i, p.Ctot - 1);    // channels past Ctot meet zero rows of the packed weights
                    bvec[(BMODE == BM_PIX) ? i : 0] = bload4(rx1, bvoff + (uint32_t)c * cs4);     // out of range + channel offset stays out of range
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    const int c = min(c0 + kc, p.Ctot - 1);
                    const bool first = c < p.C1;
                    const uint32_t soff = (uint32_t)(first ? c : c - p.C1) * cs4;      // wave-uniform: scalar offset
#pragma unroll
                    for (int i = 0; i < EPT; ++i)
                        breg[(BMODE == BM_PIX) ? 0 : kc * EPT + i] = first ? bload_s(rx1, goff[i], soff) : bload_s(rx2, goff[i], soff);
                }
            }
        };
        auto commit = [&](float* As, float* Bs, int c0) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (KT % A_RPP == 0 || tid / (BM / 4) + i * A_RPP < KT)
                    *reinterpret_cast<f32x4*>(As + (tid / (BM / 4) + i * A_RPP) * BM + 4 * (tid % (BM / 4))) = areg[i];
            if (BMODE == BM_PIX) {
#pragma unroll
                for (int i = 0; i < KCH / RPPB; ++i) {
                    const int kc = kc0 + RPPB * i;
                    f32x4 v = bvec[(BMODE == BM_PIX) ? i : 0];     // no validity select: out-of-range pixels feed discarded output columns
                    if (PRO == S2K_PRO_RELU) {
                        const int c = min(c0 + kc, p.Ctot - 1);
                        const float sc = p.bnv1[c], sh = p.bnv1[p.C1 + c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc, sh), 0.0f);
                    }
                    *reinterpret_cast<f32x4*>(Bs + kc * BN + 4 * j4) = v;
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    float sc = 1.0f, sh = 0.0f;
                    if (PRO == S2K_PRO_RELU) {       // wave-uniform channel: scalar loads
                        const int c = min(c0 + kc, p.Ctot - 1);
                        const bool first = c < p.C1;
                        sc = first ? p.bnv1[c] : p.bnv2[c - p.C1];
                        sh = first ? p.bnv1[p.C1 + c] : p.bnv2[p.C2 + c - p.C1];
                    }
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const int e = tid + NT * i;
                        float v = breg[(BMODE == BM_PIX) ? 0 : kc * EPT + i];      // padding slots were loaded as 0
                        if (PRO == S2K_PRO_RELU) v = __builtin_amdgcn_fmed3f(fmaf(v, sc, sh), 0.0f, bound[i]);   // the reference pads ACTIVATED maps
                        if (e < USED) Bs[kc * CSB + e] = v;
                    }
                }
            }
        };

        fetch(0);
        for (int ch = 0; ch < nchunks; ++ch) {
            float* As = smem + (ch & 1) * BUF;
            commit(As, As + A_FLOATS, ch * KCH);
            if (ch + 1 < nchunks) fetch((ch + 1) * KCH);
            __syncthreads();
        }
        if constexpr (TEPI) {
            __syncthreads();                                        // (E1) every wave is done with the LDS image
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) {
                __syncthreads();                                    // the consumers' tiles of pass rm are in LDS
                store_rows(rm);
                if (rm + 1 < WM) __syncthreads();                   // pass rm has been read
            }
            return;
        }
    } else {
        // =================================================================================================================
        // CONSUMER
        // =================================================================================================================
        __builtin_amdgcn_s_setprio(2);
        const int wm0 = THIN ? 0 : (wave >> 1) * (WM * 32), wn0 = (THIN ? wave : (wave & 1)) * (WN * 32);
        int boff[WN];
#pragma unroll
        for (int rn = 0; rn < WN; ++rn) {
            const int j = wn0 + rn * 32 + l31;
            if (BMODE == BM_PIX) boff[rn] = j;
            else boff[rn] = (j < R * XW) ? (j / XW) * WS + (j % XW) : 0;
        }
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        const int a_off = lh * TT * BM + wm0 + l31;                 // + (2 kk * TT + tap) * BM + rm * 32
        int b_off[WN];
#pragma unroll
        for (int rn = 0; rn < WN; ++rn) b_off[rn] = A_FLOATS + lh * CSB + boff[rn];     // + 2 kk * CSB + tap offset

        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                                        // chunk ch is in buffer ch & 1
            const float* img = smem + (ch & 1) * BUF;
            const float* Aa = img + a_off;
            const float* Bb[WN];
#pragma unroll
            for (int rn = 0; rn < WN; ++rn) Bb[rn] = img + b_off[rn];
            // step s = tap * KS + kk (compile-time after unrolling)
            auto lds_ops = [&](int s, float (&a)[WM], float (&b)[WN]) {
                const int tap = s / KS, kk = s % KS;
                const int toff = (BMODE == BM_PIX) ? 0 : (tap / 3) * WS + (tap % 3);
#pragma unroll
                for (int rm = 0; rm < WM; ++rm) a[rm] = Aa[(2 * kk * TT + tap) * BM + rm * 32];
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) b[rn] = Bb[rn][2 * kk * CSB + toff];
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
            constexpr int NS = TT * KS;
            float a0[WM], a1[WM], b0[WN], b1[WN];
            lds_ops(0, a0, b0);
#pragma unroll
            for (int s = 0; s < NS; s += 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NS) lds_ops(s + 1, a1, b1);
                mfmas(a0, b0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NS) {
                    if (s + 2 < NS) lds_ops(s + 2, a0, b0);
                    mfmas(a1, b1);
                    interleave();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---------------- epilogue (consumers hold the accumulators) ----------------------------------------------------
        bool cval[WN];
        int64_t ycol[WN];
#pragma unroll
        for (int rn = 0; rn < WN; ++rn) {
            const int j = wn0 + rn * 32 + l31;
            if (BMODE == BM_PIX) {
                const int n = nt * BN + j;
                cval[rn] = n < p.Ntot;
                const int nn = cval[rn] ? n : 0;
                const int b = nn / p.HW, pp = nn - b * p.HW;
                ycol[rn] = (int64_t)b * p.YC * HWo + pp;
            } else {
                const int r = j / XW, xx = j % XW;
                cval[rn] = (j < R * XW) && (y0 + r < p.HO) && (x0 + xx < p.WO);
                ycol[rn] = (int64_t)sb * p.YC * HWo + (int64_t)(y0 + r) * p.WO + (x0 + xx);
            }
        }
        __syncthreads();                                            // (E1) every wave is done with the LDS image
        if constexpr (TEPI) {
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rl = (THIN ? 0 : (wave >> 1) * 32) + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                    for (int rn = 0; rn < WN; ++rn) smem[rl * CTP + wn0 + rn * 32 + l31] = acc[rm][rn][reg];
                }
                __syncthreads();
                store_rows(rm);
                if (rm + 1 < WM) __syncthreads();
            }
            return;
        }
        float* srow = smem;                                         // [NWCOL n-columns of waves][2][BM]
        const int wn_idx = THIN ? wave : (wave & 1);
#pragma unroll
        for (int rm = 0; rm < WM; ++rm)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wm0 + rm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                const int gm = m0 + row;
                const bool rok = gm < p.M;
                const float bs = (p.bias && rok) ? p.bias[gm] : 0.0f;
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) {
                    float v = acc[rm][rn][reg] + bs;
                    if (rok && cval[rn]) {
                        float* dst = p.y + ycol[rn] + (int64_t)gm * HWo;
                        if (p.res) v = res_combine(v, p.res[ycol[rn] + (int64_t)gm * HWo], p.res_mul);
                        if (p.beta) v += *dst;
                        *dst = v;
                        s += v;
                        q += v * v;
                    }
                }
                if (p.stats) {
                    s = half_sum_hi(s);
                    q = half_sum_hi(q);
                    if (l31 == 31) {  // each (wave-column, row) slot has exactly one writer
                        srow[(wn_idx * 2 + 0) * BM + row] = rok ? s : 0.0f;
                        srow[(wn_idx * 2 + 1) * BM + row] = rok ? q : 0.0f;
                    }
                }
            }
    }
    // ---- both roles from here (barrier counts must match) ----
    if (producer) __syncthreads();                                  // (E1)
    if (p.stats) {
        __syncthreads();                                            // (E2) row sums are in LDS
        // partial sums of the two wave columns added in a FIXED order (BatchNorm statistics stay reproducible), then one
        // f64 atomic pair per row per workgroup into the statistics replica of this tile
        const float* srow = smem;
        double* st = p.stats + (int64_t)(tile % p.nrep) * 2 * p.M;
        for (int i = threadIdx.x; i < 2 * BM; i += 2 * NT) {
            const int row = i % BM, which = i / BM, gm = m0 + row;
            float tot = srow[(0 * 2 + which) * BM + row] + srow[(1 * 2 + which) * BM + row];
            if (NWCOL == 4) tot += srow[(2 * 2 + which) * BM + row] + srow[(3 * 2 + which) * BM + row];
            if (gm < p.M) atomic_add_d(st + which * p.M + gm, (double)tot);
        }
    }
}

// -------------------------------------------------------------------------------------------------
template <int BMODE, int WM, int KCH, int R, int XW, int PRO, bool THIN = false, bool NARROW = false>
static int launch_pc(ConvP& p, int n_ntiles, hipStream_t st) {
    constexpr int TT = (BMODE == BM_PIX) ? 1 : 9;
    constexpr int BM = THIN ? 32 : WM * 64;
    constexpr int BNP = NARROW ? 64 : 128;
    constexpr int USED = (R + 2) * (XW + 2);
    constexpr int CSB = (BMODE == BM_PIX) ? BNP : USED;
    constexpr int BUF = (KCH * TT * BM + KCH * CSB + 3) & ~3;
    constexpr size_t ct_bytes = (BMODE == BM_PIX || XW % 4 == 0) ? (size_t)(THIN ? 32 : 64) * ((THIN ? 256 : BNP) + 8) * sizeof(float) : 0;   // transposed store
    constexpr size_t lds = std::max((size_t)2 * BUF * sizeof(float), ct_bytes);
    static_assert(lds <= 160 * 1024, "LDS image");
    static_assert((THIN ? 8 : 4) * BM * sizeof(float) <= lds, "statistics rows fit in the image");
    p.n_mtiles = cdiv(p.M, BM);
    {   // 32-bit buffer offsets: image-local tiles need one image < 2 GiB, tiles that may straddle images the whole tensor
        // (descriptors are based at the first image a tile touches)
        const int64_t span = (BMODE == BM_SPATIAL || (p.HW % BNP) == 0) ? 1 : std::min<int64_t>(p.B, (BNP - 2) / p.HW + 2);
        const int64_t img1 = (int64_t)p.C1 * p.H * p.W * 4, img2 = (int64_t)p.C2 * p.H * p.W * 4;
        const int64_t need = std::max(img1, img2) * span;
        if (need >= 0x7ffffff0ll) { set_error("conv: the %lld image(s) one tile touches exceed 2 GiB (%lld B)", (long long)span, (long long)need); return S2K_EINVAL; }
    }
    const int64_t blocks = (int64_t)p.n_mtiles * n_ntiles;
    if (blocks <= 0 || blocks > 0x7fffffff) { set_error("conv: bad grid %lld", (long long)blocks); return S2K_EINVAL; }
    p.n_tiles = (int)blocks;
    p.splits = 1;
    auto kern = conv_pc_kernel<BMODE, WM, KCH, R, XW, PRO, THIN, NARROW>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, p);
    g_s2k_variant = 1;
    return S2K_OK;
}

template <int BMODE, int KCH, int R, int XW>
static int launch_pc_bm(ConvP& p, int n_ntiles, int bm, hipStream_t st) {
    const bool relu = p.pro1 == S2K_PRO_RELU;
    if (bm == 128) return relu ? launch_pc<BMODE, 2, KCH, R, XW, S2K_PRO_RELU>(p, n_ntiles, st) : launch_pc<BMODE, 2, KCH, R, XW, S2K_PRO_NONE>(p, n_ntiles, st);
    return relu ? launch_pc<BMODE, 1, KCH, R, XW, S2K_PRO_RELU>(p, n_ntiles, st) : launch_pc<BMODE, 1, KCH, R, XW, S2K_PRO_NONE>(p, n_ntiles, st);
}

int launch_conv_pc(ConvP& p, hipStream_t st) {
    static const int enabled = tune_int("S2K_CONV_PC", 3);         // bit 0: 1x1, bit 1: 3x3
    if (p.mode != S2K_MODE_CONV || p.S != 1 || p.gate1) return 1;
    if (p.pro1 != S2K_PRO_NONE && p.pro1 != S2K_PRO_RELU) return 1;
    if (p.C2 > 0 && p.pro2 != p.pro1) return 1;
    if (p.M <= 32 && p.KH == 3 && p.KW == 3) {                        // thin 3x3: one 32-row m-tile, 4 x 64 pixel tiles
        static const int thin = tune_int("S2K_CONV_PC_THIN", 1);
        if (!thin || p.HO != p.H || p.WO != p.W || p.PT != 1 || p.PL != 1 || p.WO % 64 != 0 || p.M < 20) return 1;
        p.R = 4; p.XW = 64; p.IR = 6; p.IC = 66; p.WS = 66; p.CS = 6 * 66;
        p.tiles_x = p.WO / 64;
        p.tiles_y = cdiv(p.HO, 4);
        const int n = p.B * p.tiles_x * p.tiles_y;
        if (n < 512) return 1;
        static const int kch = tune_int("S2K_CONV_PC_THIN_KCH", 4);
        const bool relu = p.pro1 == S2K_PRO_RELU;
        if (kch == 8) return relu ? launch_pc<BM_SPATIAL, 1, 8, 4, 64, S2K_PRO_RELU, true>(p, n, st) : launch_pc<BM_SPATIAL, 1, 8, 4, 64, S2K_PRO_NONE, true>(p, n, st);
        return relu ? launch_pc<BM_SPATIAL, 1, 4, 4, 64, S2K_PRO_RELU, true>(p, n, st) : launch_pc<BM_SPATIAL, 1, 4, 4, 64, S2K_PRO_NONE, true>(p, n, st);
    }
    if (p.M < 48) return 1;                                           // thin layers: the wide-pixel tiles of igemm.hip
    // tile height: 128 unless it pads M by more than 12 %
    const int bm = ((double)cdiv(p.M, 128) * 128 / p.M <= 1.12) ? 128 : 64;
    const int T = p.KH * p.KW;
    if (T == 1) {
        if (!(enabled & 1) || p.C2 != 0 || (p.HW & 3) || p.HO != p.H || p.WO != p.W) return 1;
        const int n_ntiles = cdiv(p.Ntot, 128);
        if ((int64_t)cdiv(p.M, bm) * n_ntiles < 192) {
            // too few 128-pixel tiles for 256 CUs: 64 x 64 tiles when they fill the chip and the reduction is long enough to
            // amortise their epilogue, else the split-K path of igemm.hip
            static const int narrow = tune_int("S2K_CONV_PC_NARROW", 1);
            const int n64 = cdiv(p.Ntot, 64);
            if (!narrow || p.Ctot < 512 || (int64_t)cdiv(p.M, 64) * n64 < 384) return 1;
            return p.pro1 == S2K_PRO_RELU ? launch_pc<BM_PIX, 1, 32, 1, 64, S2K_PRO_RELU, false, true>(p, n64, st)
                                          : launch_pc<BM_PIX, 1, 32, 1, 64, S2K_PRO_NONE, false, true>(p, n64, st);
        }
        if (p.Ctot < 256) return 1;                                     // short reductions have nothing to pipeline: generic kernels
        static const int kch1 = tune_int("S2K_CONV_PC_KCH1", 32);      // 32: two workgroups per CU (64 KB of LDS each)
        if (kch1 == 64) return launch_pc_bm<BM_PIX, 64, 1, 64>(p, n_ntiles, bm, st);
        return launch_pc_bm<BM_PIX, 32, 1, 64>(p, n_ntiles, bm, st);
    }
    if (!(enabled & 2) || T != 9 || p.KH != 3 || p.HO != p.H || p.WO != p.W || p.PT != 1 || p.PL != 1) return 1;
    auto tiles = [&](int r, int xw) {
        p.R = r; p.XW = xw; p.IR = r + 2; p.IC = xw + 2; p.WS = xw + 2; p.CS = p.IR * p.WS;
        p.tiles_x = cdiv(p.WO, xw);
        p.tiles_y = cdiv(p.HO, r);
        return p.B * p.tiles_x * p.tiles_y;
    };
    // chunk of 4 channels: 45 KB of LDS, two to three workgroups per CU hide each other's pipeline fill and epilogue (M = 128:
    // 112 vs 105 TF/s); with >= 2 m-tiles of 128 rows the activation tile is shared through L2 and the 8-channel chunk wins
    // (M = 512: 124 vs 118 TF/s)
    static const int kch3 = tune_int("S2K_CONV_PC_KCH3", 4);
#define PC3(RR, XX) { const int n = tiles(RR, XX); if ((int64_t)cdiv(p.M, bm) * n < 160) return 1; \
                      if (kch3 == 8 || (kch3 == 4 && p.M >= 256)) return launch_pc_bm<BM_SPATIAL, 8, RR, XX>(p, n, bm, st); \
                      return launch_pc_bm<BM_SPATIAL, 4, RR, XX>(p, n, bm, st); }
    if (p.WO >= 64 && p.WO % 64 == 0) PC3(2, 64)
    if (p.WO == 32) PC3(4, 32)
    if (p.WO == 16) PC3(8, 16)
    if (p.WO == 56 || p.WO == 112 || p.WO == 224) PC3(2, 56)
    if (p.WO == 28) PC3(4, 28)
    if (p.WO == 14) PC3(8, 14)
#undef PC3
    return 1;
}

}  // namespace s2k
