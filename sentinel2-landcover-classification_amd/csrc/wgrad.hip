// Weight gradients of the dense convs on the gfx950 f32 matrix cores; K = pixels.
//
// Replaces ATen convolution_backward's grad_weight for nn.Conv2d 1x1 / 3x3 / stem and
// nn.ConvTranspose2d k2 s2 (reference efficientnet_unet.py:114-121,168-176,191-197,319-372):
//   conv :  dW[tap][m][c] += sum_pix dY[b][m][yo][xo]      * Xpro[b][c][yo*S+ky-PT][xo*S+kx-PL]
//   convT:  dW[tap][m][c] += sum_pix Xpro[b][m][y][x]      * G[b][c][2y+dy][2x+dx]
// MFMA 32x32x2 with k = two consecutive pixels: lane l holds P[m = l&31][pix + (l>>5)] and
// Q[c = l&31][pix + (l>>5) (+tap shift)]; both LDS tiles are [channel][pixels] with an ODD channel
// stride, so the 32 lanes of a half-wave hit 32 different banks.  One accumulator tile per tap; the
// activations are re-normalised/activated on load (same prologue as the forward), never stored.
// Partial sums of the pixel splits are combined with float atomics into a [T][M][C] scratch whose
// rows are contiguous along c (128-B segments per half-wave: the full-rate atomic shape).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "wgrad.h"

namespace s2k {

__device__ __forceinline__ float ld_pro(const float* x, const float* bnv, const float* gate, int pro, int C, int c,
                                        int64_t off, int gate_row) {
    float v = x[off];
    if (pro != S2K_PRO_NONE) v = apply_pro(v, pro, bnv[c], bnv[C + c]);
    if (gate) v *= gate[gate_row + c];
    return v;
}

// T taps, per-wave tile (WM*32) x (WN*32), waves arranged WVM x WVN x WVK (K = pixel pairs).
// NPJ = pixel slots per tile (power of two >= pixels per tile), EPT = halo elements per thread per
// channel.  A thread owns fixed (pixel | halo element) slots, only the tile origin changes, so no
// index arithmetic is left in the tile loop; the operands of tile i+1 are fetched into registers
// before the MFMAs of tile i are issued and written to LDS after them (latency hidden under
// 72..288 MFMAs per wave).
// WSC: compile-time row stride of the Q halo tile (0 = run time).  With it the 9 tap offsets of the operand reads are
// instruction immediates: the pair loop needs one address add instead of ten.
template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK, int NPJ, int EPT, int PROP, int PROQ, int WSC = 0>
__global__ void __launch_bounds__(NTHREADS) wgrad_kernel(const WgradP p) {
    constexpr int BM = WM * WVM * 32;
    constexpr int BC = WN * WVN * 32;
    constexpr int RSTEP = NTHREADS / NPJ;   // P/Q rows staged per pass
    constexpr int NPR = BM / RSTEP;         // P values per thread per tile
    constexpr int NQR = (MODE == WG_SPATIAL) ? BC * EPT : (MODE == WG_GATHER ? 4 * (BC / RSTEP) : BC / RSTEP);
    static_assert(WVM * WVN * WVK == 4, "4 waves per workgroup");
    static_assert(NTHREADS % NPJ == 0 && BM % RSTEP == 0 && BC % RSTEP == 0, "pixel slots vs threads");
    // SPATIAL: compile-time LDS row strides (tap / channel offsets of the operand reads and of the stager's stores fold
    // into instruction immediates instead of vector adds — the f32 MFMA holds the SIMD's vector issue port, so every
    // VALU instruction costs MFMA time)
    constexpr int PSTR = NPJ + 1;
    constexpr int CSQ = (MODE == WG_SPATIAL) ? NTHREADS * EPT + 1 : (MODE == WG_GATHER ? 4 * NPJ + 1 : NPJ + 1);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ps = smem;                    // [BM][PSTR]
    float* Qs = smem + BM * PSTR;        // [BC][CSQ]
    float* psc = Qs + BC * CSQ;          // BatchNorm scale/shift of this workgroup's P rows / Q columns,
    float* psh = psc + BM;               // staged once so that the commit step has no loads behind branches
    float* qsc = psh + BM;
    float* qsh = qsc + BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wk = wave % WVK;
    const int wmn = wave / WVK;
    const int wm0 = (wmn / WVN) * (WM * 32);
    const int wc0 = (wmn % WVN) * (WN * 32);
    const int mt = blockIdx.x % p.n_mtiles;
    const int ct = blockIdx.x / p.n_mtiles;
    const int m0 = mt * BM, c0 = ct * BC;
    const int tile_begin = blockIdx.y * p.tiles_per_split;
    int tile_end = tile_begin + p.tiles_per_split;
    if (tile_end > p.ntiles) tile_end = p.ntiles;

    // ---- fixed per-thread stager slots --------------------------------------------------------------
    const int pj = tid % NPJ;            // pixel slot inside the tile
    const int prow = tid / NPJ;          // first P/Q row of this thread
    int pr = 0, pxx = 0;                 // SPATIAL: (row, col) of the slot inside the tile
    int qrr[EPT], qcc[EPT];
    if (MODE == WG_SPATIAL) {
        pr = pj / p.XWe;
        pxx = pj - pr * p.XWe;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + NTHREADS * i;
            qrr[i] = e / p.WS;
            qcc[i] = e - qrr[i] * p.WS;
        }
    }
    const int np_sp = p.R * p.XWe;
    const int used_sp = p.IR * p.WS;
    const int half_xw = p.XWe >> 1;
    const int npairs = (MODE == WG_SPATIAL) ? (np_sp >> 1) : (NPJ >> 1);

    f32x16 acc[T][WM][WN];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.0f;

    // Operand loads are unconditional bounds-checked buffer loads (validity is a select afterwards): one
    // affine 32-bit offset stream per operand, no branch around any load
    // Descriptors are based at the image of the fetched tile whenever a pixel tile cannot straddle two images (always for
    // the 3x3 tiles; H*W % NPJ == 0 otherwise), so offsets stay image-local and tensors above 2 GiB are addressable.
    const bool img_local = (MODE == WG_SPATIAL) || (p.HWp % NPJ) == 0;
    rsrc_t rp = make_rsrc(p.p, (int64_t)p.B * p.M * p.HWp * 4);
    rsrc_t rq = make_rsrc(p.q, (int64_t)p.B * p.C * p.HWq * 4);
    const rsrc_t rg = make_rsrc(p.gateq ? p.gateq : p.q, p.gateq ? (int64_t)p.B * p.C * 4 : 0);
    const uint32_t p_rstride = (uint32_t)RSTEP * p.HWp * 4u, q_rstride = (uint32_t)RSTEP * p.HWq * 4u;
    float preg[NPR], qreg[NQR];
    float qgate[(MODE == WG_PIX) ? BC / RSTEP : 1];   // SE gate of the fetched Q elements (1x1 project conv)
    for (int i = tid; i < BM; i += NTHREADS) {
        const int gm = m0 + i;
        const bool on = p.prop != S2K_PRO_NONE && gm < p.M;
        psc[i] = on ? p.bnvp[gm] : 1.0f;
        psh[i] = on ? p.bnvp[p.M + gm] : 0.0f;
    }
    for (int i = tid; i < BC; i += NTHREADS) {
        const int gc = c0 + i;
        const bool on = p.proq != S2K_PRO_NONE && gc < p.C;
        qsc[i] = on ? p.bnvq[gc] : 1.0f;
        qsh[i] = on ? p.bnvq[p.C + gc] : 0.0f;
    }
    bool f_pok = false;           // the fetched tile's pixel slot is inside the image
    unsigned f_qok = 0;           // SPATIAL: bit i = halo slot i inside the image
    int f_b = 0;                  // image index of the fetched tile (gate rows)

    // ---------------- global -> registers (raw values) ---------------------------------------------------
    auto fetch = [&](int tile) {
        if (MODE == WG_SPATIAL) {
            const int tx = tile % p.tiles_x;
            const int ty = (tile / p.tiles_x) % p.tiles_y;
            const int b = tile / (p.tiles_x * p.tiles_y);
            const int y0 = ty * p.R, x0 = tx * p.XW;
            const int yo = y0 + pr, xo = x0 + pxx;
            f_b = b;
            rp = make_rsrc(p.p + (int64_t)b * p.M * p.HWp, (int64_t)p.M * p.HWp * 4);
            rq = make_rsrc(p.q + (int64_t)b * p.C * p.HWq, (int64_t)p.C * p.HWq * 4);
            f_pok = pj < np_sp && pxx < p.XW && yo < p.HO && xo < p.WO;
            // per-lane part (pixel, or out of range) in the vector offset, per-row part in the SCALAR offset: no vector
            // arithmetic per load.  Rows / channels past the tensor are clamped (their products land in discarded outputs).
            const uint32_t pvoff = f_pok ? (uint32_t)(yo * p.WO + xo) * 4u : BUF_OOB;
            const int prow_u = __builtin_amdgcn_readfirstlane(prow);   // NPJ >= 64: one P row per wave and pass
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                const int row = min(m0 + prow_u + i * RSTEP, p.M - 1);
                preg[i] = bload_s(rp, pvoff, (uint32_t)(row * p.HWp) * 4u);
            }
            const int iy0 = y0 * p.S - p.PT, ix0 = x0 * p.S - p.PL;
            uint32_t goff[EPT];
            f_qok = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int iy = iy0 + qrr[i], ix = ix0 + qcc[i];
                const bool ok = (tid + NTHREADS * i) < used_sp && qcc[i] < p.IC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                goff[i] = ok ? (uint32_t)(iy * p.W + ix) * 4u : BUF_OOB;
                f_qok |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int c = 0; c < BC; ++c) {
                const uint32_t soff = (uint32_t)(min(c0 + c, p.C - 1) * p.HWq) * 4u;
#pragma unroll
                for (int i = 0; i < EPT; ++i) qreg[c * EPT + i] = bload_s(rq, goff[i], soff);
            }
        } else {
            const int64_t ntot = (int64_t)p.B * p.HWp;
            const int64_t n = (int64_t)tile * NPJ + pj;
            f_pok = n < ntot;
            const int64_t nn = f_pok ? n : 0;
            const int b = (int)(nn / p.HWp);
            const int pp = (int)(nn - (int64_t)b * p.HWp);
            f_b = b;
            // descriptors based at the first image this pixel tile touches (uniform): offsets span the tile's images only
            const int bt = (int)(((int64_t)tile * NPJ) / p.HWp);
            rp = make_rsrc(p.p + (int64_t)bt * p.M * p.HWp, (int64_t)(img_local ? 1 : p.B - bt) * p.M * p.HWp * 4);
            rq = make_rsrc(p.q + (int64_t)bt * p.C * p.HWq, (int64_t)(img_local ? 1 : p.B - bt) * p.C * p.HWq * 4);
            const int brel = b - bt;
            // per-lane part (image, pixel; or out of range) in the vector offset, the row / channel part in the scalar
            // offset (clamped into the tensor: rows and channels past the end only feed discarded outputs)
            const int prow_u = __builtin_amdgcn_readfirstlane(prow);
            const uint32_t pvoff = f_pok ? (uint32_t)((int64_t)brel * p.M * p.HWp + pp) * 4u : BUF_OOB;
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                if constexpr (NPJ >= 64) {   // one P row per wave and pass: the row offset is scalar
                    preg[i] = bload_s(rp, pvoff, (uint32_t)(min(m0 + prow_u + i * RSTEP, p.M - 1) * p.HWp) * 4u);
                } else {
                    const int row = m0 + prow + i * RSTEP;
                    preg[i] = bload(rp, (f_pok && row < p.M) ? pvoff + (uint32_t)(row * p.HWp) * 4u : BUF_OOB);
                }
            }
            if (MODE == WG_PIX) {
                const uint32_t qvoff = f_pok ? (uint32_t)((int64_t)brel * p.C * p.HWq + pp) * 4u : BUF_OOB;
                const uint32_t gvoff = f_pok ? (uint32_t)(b * p.C) * 4u : BUF_OOB;
#pragma unroll
                for (int i = 0; i < BC / RSTEP; ++i) {
                    const int ch = min(c0 + prow_u + i * RSTEP, p.C - 1);
                    qreg[i] = bload_s(rq, qvoff, (uint32_t)(ch * p.HWq) * 4u);
                    qgate[(MODE == WG_PIX) ? i : 0] = bload_s(rg, gvoff, (uint32_t)ch * 4u);  // 0-size descriptor without a gate -> 0
                }
            } else {  // GATHER: Q is [B][C][2HO][2WO]
                const int yy = pp / p.WO, xx = pp - yy * p.WO;
                const uint32_t qbase = (uint32_t)(((int64_t)brel * p.C + c0 + prow) * p.HWq + (int64_t)(2 * yy) * p.W + 2 * xx) * 4u;
#pragma unroll
                for (int i = 0; i < BC / RSTEP; ++i) {
                    const bool ok = f_pok && c0 + prow + i * RSTEP < p.C;
                    const uint32_t o = ok ? qbase + i * q_rstride : BUF_OOB, w4 = ok ? (uint32_t)p.W * 4u : 0u;
                    qreg[4 * i + 0] = bload(rq, o); qreg[4 * i + 1] = bload(rq, o + 4u);
                    qreg[4 * i + 2] = bload(rq, o + w4); qreg[4 * i + 3] = bload(rq, o + w4 + 4u);
                }
            }
        }
    };

    // ---------------- registers -> LDS (prologues applied here; zero padding stays zero) ----------------
    // branch-free per element: parameters from LDS, validity as a final select
    auto commit = [&]() {
        if (MODE == WG_SPATIAL) {        // out-of-image pixels were loaded as 0 (no prologue on this operand)
            static_assert(MODE != WG_SPATIAL || PROP == S2K_PRO_NONE, "3x3 wgrad: P (= dY) carries no prologue");
            if (pj < np_sp) {
#pragma unroll
                for (int i = 0; i < NPR; ++i) Ps[(prow + i * RSTEP) * PSTR + pj] = preg[i];
            }
        } else {
            const float keep = f_pok ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
                const int m = prow + i * RSTEP;
                float v = preg[i];       // out-of-range pixels were loaded as 0
                if (PROP != S2K_PRO_NONE) v = apply_pro_c<PROP>(v, psc[m], psh[m]) * keep;
                Ps[m * PSTR + pj] = v;
            }
        }
        if (MODE == WG_GATHER) {
#pragma unroll
            for (int i = 0; i < BC / RSTEP; ++i) {
                float* dst = Qs + (prow + i * RSTEP) * CSQ + pj;
                dst[0] = qreg[4 * i + 0];
                dst[NPJ] = qreg[4 * i + 1];
                dst[2 * NPJ] = qreg[4 * i + 2];
                dst[3 * NPJ] = qreg[4 * i + 3];
            }
        } else if (MODE == WG_SPATIAL) {
            // branch-free and address-arithmetic-free: the scale/shift table is read through ONE laundered LDS base
            // register + immediates, padding slots are zeroed by a 0/1 multiplier (hipcc turns `ok ? pro(v) : 0` into a
            // branch per element and re-materialises every constant LDS address with a v_mov)
            const float* qtab = qsc;
            asm volatile("" : "+v"(qtab));
            float keepf[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) keepf[i] = ((f_qok >> i) & 1u) ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < BC; ++c) {
                float sc = 1.0f, sh = 0.0f;
                if (PROQ != S2K_PRO_NONE) { sc = qtab[c]; sh = qtab[BC + c]; }
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int e = tid + NTHREADS * i;
                    float v = qreg[c * EPT + i];       // padding / out-of-tile slots were loaded as 0
                    if (PROQ != S2K_PRO_NONE) v = apply_pro_c<PROQ>(v, sc, sh) * keepf[i];
                    if (e < used_sp) Qs[c * CSQ + e] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BC / RSTEP; ++i) {
                const int c = prow + i * RSTEP;
                float v = qreg[i];       // no validity select: P is 0 wherever the pixel is out of range, and
                if (PROQ != S2K_PRO_NONE) v = apply_pro_c<PROQ>(v, qsc[c], qsh[c]);   // channels past C feed discarded columns
                if (p.gateq) v *= qgate[(MODE == WG_PIX) ? i : 0];
                Qs[c * CSQ + pj] = v;
            }
        }
    };

    __syncthreads();  // scale/shift staged
    if (tile_begin < tile_end) {
        fetch(tile_begin);
        commit();
    }
    __syncthreads();

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool more = tile + 1 < tile_end && !(p.exp & 1);
        if (more) fetch(tile + 1);
        // ---------------- MFMA over pixel pairs ----------------------------------------------------
        // SPATIAL: pair s = (row r, column pair xp); walk (r, xp) incrementally, WVK pairs at a time.
        // Two operand sets: the LDS reads of pair s+WVK are issued before the T MFMAs of pair s (hipcc left to
        // itself re-uses ONE register for every B read and waits for each read in front of its MFMA).
        int r_run = 0, xp_run = wk;
        if (MODE == WG_SPATIAL)
            while (xp_run >= half_xw) { xp_run -= half_xw; ++r_run; }
        auto lds_operands = [&](int s, float (&a)[WM], float (&bq)[T][WN]) {
            const bool live = s < npairs;          // past the end: re-read pair 0 (never used), keeps the loop branch-free
            const int n = live ? 2 * s + lh : lh;  // this lane's pixel inside the tile
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = Ps[(wm0 + rm * 32 + l31) * PSTR + n];
            int qbase;
            if (MODE == WG_SPATIAL) {
                qbase = live ? (r_run * p.S) * (WSC ? WSC : p.WS) + (2 * xp_run + lh) * p.S : 0;
                xp_run += WVK;
                while (xp_run >= half_xw) { xp_run -= half_xw; ++r_run; }
            } else {
                qbase = n;
            }
            int tdy = 0, tdx = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int toff;
                if (MODE == WG_SPATIAL) toff = WSC ? (t / 3) * WSC + (t % 3) : tdy * p.WS + tdx;   // T == 9: 3x3 taps
                else if (MODE == WG_GATHER) toff = t * NPJ;
                else toff = 0;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) bq[t][rn] = Qs[(wc0 + rn * 32 + l31) * CSQ + qbase + toff];
                if (++tdx == p.KW) { tdx = 0; ++tdy; }
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
        float a0[WM], a1[WM], b0[T][WN], b1[T][WN];
        // issue order inside one half-iteration: MFMA, then a slice of the NEXT pair's address arithmetic and LDS reads,
        // MFMA, ... — the 64-cycle MFMA leaves its SIMD's issue port free for ~48 cycles, which is where that work goes
        auto interleave = [&]() {
#pragma unroll
            for (int t = 0; t < T * WM * WN; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // <= 2 DS reads
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // <= 4 VALU / SALU
            }
        };
        lds_operands(wk, a0, b0);
        for (int s = wk; s < ((p.exp & 4) ? 0 : npairs); s += 2 * WVK) {
            __builtin_amdgcn_sched_barrier(0);
            lds_operands(s + WVK, a1, b1);
            mfmas(a0, b0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (s + WVK >= npairs) break;
            lds_operands(s + 2 * WVK, a0, b0);
            mfmas(a1, b1);
            interleave();
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (more) {
            commit();
            __syncthreads();
        }
    }

    // ---------------- combine: wgs[t][m][c] += acc (k-waves reduced through LDS first: wgrad.h) -------
    if (p.exp & 2) return;
    if (tile_begin >= tile_end) return;
    wg_combine<T, WM, WN, WVK>(p, acc, smem, wk, wmn, lane, m0, c0, wm0, wc0);     // (the loop's last barrier freed the tile images)
}

// -------------------------------------------------------------------------------------------------
template <typename T>
static T* ref_ptr(const Ctx& c, int64_t ref) {
    if (ref < 0) return nullptr;
    const int base = (int)(ref >> 56);
    const int64_t off = ref & ((1ll << 56) - 1);
    if (base >= c.n_bases || c.bases[base] == nullptr) return reinterpret_cast<T*>(1);
    return reinterpret_cast<T*>(static_cast<char*>(c.bases[base]) + off);
}

template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK, int NPJ, int EPT, int PROP, int PROQ, int WSC = 0>
static int launch_wg2(WgradP& p, hipStream_t st) {
    constexpr int BM = WM * WVM * 32, BC = WN * WVN * 32;
    p.n_mtiles = cdiv(p.M, BM);
    p.n_ctiles = cdiv(p.C, BC);
    p.PSTR = NPJ + 1;                                                               // compile-time strides in the kernel
    p.CSQ = (MODE == WG_SPATIAL) ? NTHREADS * EPT + 1 : (MODE == WG_GATHER ? 4 * NPJ + 1 : NPJ + 1);
    {   // 32-bit buffer offsets: one image must stay below 2 GiB when tiles are image-local, else the whole tensor
        const int64_t span = ((MODE == WG_SPATIAL) || (p.HWp % NPJ) == 0) ? 1 : std::min<int64_t>(p.B, (NPJ - 2) / p.HWp + 2);   // images one pixel tile touches
        const int64_t need = std::max((int64_t)p.M * p.HWp, (int64_t)p.C * p.HWq) * 4 * span;
        if (need >= 0x7ffffff0ll) { set_error("wgrad: the %lld image(s) one pixel tile touches exceed 2 GiB", (long long)span); return S2K_EINVAL; }
    }
    const size_t lds = ((size_t)BM * p.PSTR + (size_t)BC * p.CSQ + 2 * (BM + BC)) * sizeof(float);
    if (p.gatep || (p.gateq && MODE != WG_PIX)) { set_error("wgrad: SE gate is only supported on the Q operand of 1x1 convs"); return S2K_EINVAL; }
    if (lds > 160 * 1024) { set_error("wgrad: LDS %zu too large", lds); return S2K_EINVAL; }
    auto kern = wgrad_kernel<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, PROP, PROQ, WSC>;
    if (MODE == WG_SPATIAL && p.IR * p.WS > NTHREADS * EPT) { set_error("wgrad: halo exceeds EPT"); return S2K_EINVAL; }
    if (MODE == WG_SPATIAL && p.R * p.XWe > NPJ) { set_error("wgrad: tile exceeds pixel slots"); return S2K_EINVAL; }
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // Pixel splits.  The 9-tap kernels hold 144 accumulator registers (one workgroup per CU), the 1x1 / 2x2 kernels run
    // ~3 per CU: pick the split count whose workgroup count fills whole rounds of those slots (288 workgroups on 256
    // slots take two rounds: 1.8x the time of 256), preferring fewer splits (each ends in an atomic combine of its tile).
    const int mc = p.n_mtiles * p.n_ctiles;
    // workgroups that run at once: the 9-tap and the 128x128 kernels need > 256 registers (one workgroup per CU)
    const int slots = (T == 9 || WM * WN >= 4) ? 256 : (WM * WN >= 2 ? 512 : 768);
    int max_splits = cdiv(p.ntiles, 4);          // at least 4 pixel tiles per split
    if (max_splits > 65535) max_splits = 65535;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 1e30;
    const int s_hi = std::min(max_splits, std::max(1, 4 * slots / mc));
    for (int sp = 1; sp <= s_hi; ++sp) {
        const double rounds = (double)cdiv(mc * sp, slots);
        // tiles + pipeline fill + the atomic combine of the accumulator tile, in tile times
        // (a single round has no slack for a slow workgroup: a few rounds of shorter workgroups balance better)
        const double cost = rounds * ((double)cdiv(p.ntiles, sp) + (T == 9 ? 1.5 : (WM * WN >= 4 ? 2.0 : 0.7))) * (T == 9 ? 1.0 : 1.0 + 0.15 / rounds);
        if (cost < best * 0.999) { best = cost; splits = sp; }
    }
    p.tiles_per_split = cdiv(p.ntiles, splits);
    splits = cdiv(p.ntiles, p.tiles_per_split);
    hipLaunchKernelGGL(kern, dim3(mc, splits), dim3(NTHREADS), lds, st, p);
    return S2K_OK;
}

// the prologue kinds are compile-time in the kernel (no per-element branches in the LDS commit); only the
// combinations the planners emit are instantiated
template <int MODE, int T, int WM, int WN, int WVM, int WVN, int WVK, int NPJ, int EPT, int WSC = 0>
static int launch_wg(WgradP& p, hipStream_t st) {
    const int pp = p.prop, pq = p.proq;
    if constexpr (WSC != 0) {
        if (pp == S2K_PRO_NONE && pq == S2K_PRO_NONE) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_NONE, WSC>(p, st);
        if (pp == S2K_PRO_NONE && pq == S2K_PRO_RELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_RELU, WSC>(p, st);
    }
    if constexpr (MODE == WG_GATHER) {
        if (pq == S2K_PRO_NONE && pp == S2K_PRO_RELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_RELU, S2K_PRO_NONE>(p, st);
        if (pq == S2K_PRO_NONE && pp == S2K_PRO_SILU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_SILU, S2K_PRO_NONE>(p, st);
        if (pq == S2K_PRO_NONE && pp == S2K_PRO_NONE) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
        if (pq == S2K_PRO_NONE && pp == S2K_PRO_GELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_GELU, S2K_PRO_NONE>(p, st);
    } else {
        if (pp == S2K_PRO_NONE && pq == S2K_PRO_NONE) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_NONE>(p, st);
        if (pp == S2K_PRO_NONE && pq == S2K_PRO_RELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_RELU>(p, st);
        if constexpr (MODE == WG_PIX)
            if (pp == S2K_PRO_NONE && pq == S2K_PRO_SILU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_SILU>(p, st);
        if constexpr (MODE == WG_PIX)
            if (pp == S2K_PRO_NONE && pq == S2K_PRO_GELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_NONE, S2K_PRO_GELU>(p, st);
        // ConvTranspose weight gradients as 1x1 contractions over the space-to-depth gradient: the prologue sits on P (the layer input)
        if constexpr (MODE == WG_PIX) {
            if (pq == S2K_PRO_NONE && pp == S2K_PRO_RELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_RELU, S2K_PRO_NONE>(p, st);
            if (pq == S2K_PRO_NONE && pp == S2K_PRO_SILU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_SILU, S2K_PRO_NONE>(p, st);
            if (pq == S2K_PRO_NONE && pp == S2K_PRO_GELU) return launch_wg2<MODE, T, WM, WN, WVM, WVN, WVK, NPJ, EPT, S2K_PRO_GELU, S2K_PRO_NONE>(p, st);
        }
    }
    set_error("wgrad: prologue combination (P %d, Q %d) is not instantiated for this mode", pp, pq);
    return S2K_EINVAL;
}

int launch_wgrad(const S2kOp& op, const Ctx& c) {
    WgradP p{};
    p.p = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_P]);
    p.bnvp = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_BNVP]);
    p.gatep = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_GATEP]);
    p.q = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_Q]);
    p.bnvq = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_BNVQ]);
    p.gateq = ref_ptr<const float>(c, op.t[S2K_WGRAD_T_GATEQ]);
    p.wgs = ref_ptr<float>(c, op.t[S2K_WGRAD_T_WGS]);
    const void* ptrs[] = {p.p, p.bnvp, p.gatep, p.q, p.bnvq, p.gateq, p.wgs};
    for (const void* q : ptrs)
        if (q == reinterpret_cast<const void*>(1)) { set_error("wgrad: tensor references a null base"); return S2K_EFAULT; }
    const int32_t* d = op.d;
    p.B = d[S2K_WGRAD_D_B]; p.M = d[S2K_WGRAD_D_M]; p.C = d[S2K_WGRAD_D_C]; p.CTOT = d[S2K_WGRAD_D_CTOT];
    p.H = d[S2K_WGRAD_D_H]; p.W = d[S2K_WGRAD_D_W]; p.KH = d[S2K_WGRAD_D_KH]; p.KW = d[S2K_WGRAD_D_KW];
    p.S = d[S2K_WGRAD_D_STRIDE]; p.PT = d[S2K_WGRAD_D_PAD_T]; p.PL = d[S2K_WGRAD_D_PAD_L];
    p.HO = d[S2K_WGRAD_D_HO]; p.WO = d[S2K_WGRAD_D_WO]; p.prop = d[S2K_WGRAD_D_PROP]; p.proq = d[S2K_WGRAD_D_PROQ];
    const int mode = d[S2K_WGRAD_D_MODE];
    p.p_bf16 = d[S2K_WGRAD_D_P_BF16];
    p.T = p.KH * p.KW;
    p.HWp = p.HO * p.WO;
    p.HWq = p.H * p.W;
    if (!p.p || !p.q || !p.wgs || p.B <= 0 || p.M <= 0 || p.C <= 0) { set_error("wgrad: missing tensor / bad dims"); return S2K_EINVAL; }
    if ((p.prop != S2K_PRO_NONE && !p.bnvp) || (p.proq != S2K_PRO_NONE && !p.bnvq)) {
        set_error("wgrad: prologue without BNV"); return S2K_EINVAL;
    }
    p.R = p.XW = p.XWe = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = 0;
    static const int wg_exp = tune_int("S2K_WG_EXP", 0);
    p.exp = wg_exp;
    hipStream_t st = c.stream;
    const int64_t npix = (int64_t)p.B * p.HWp;
    if (op.flags & S2K_FLAG_BF16) {   // bf16-mixed plan: the shapes of wgrad_bf16.hip round their MFMA operands to bf16; 1 = not one of them
        const int rc = launch_wgrad_bf16(p, mode, st);
        if (rc != 1) return rc;
        p.R = p.XW = p.XWe = p.tiles_x = p.tiles_y = p.IR = p.IC = p.WS = 0;
    }
    if (p.p_bf16) {   // planned only where wgrad_bf16.hip takes the stage (plan/bf16.py); the f32 kernels would read the halves as floats
        set_error("wgrad: P_BF16 on a stage the bf16 1x1 kernel does not take (FLAG_BF16 missing or shape not in its list)");
        return S2K_EINVAL;
    }
    {   // 1x1 / Linear weight gradients whose pixel rows are 16-byte multiples: the quad-read producer / consumer kernel (wgrad_q4.hip)
        const int rc = launch_wgrad_q4(p, mode, st);
        if (rc != 1) return rc;
    }
    {   // the MFMA-bound shapes run on the producer / consumer kernels (wgrad_pc.hip); 1 = not one of theirs
        const int rc = launch_wgrad_pc(p, mode, st);
        if (rc != 1) return rc;
    }

    if (mode == S2K_MODE_GATHER2X2) {
        if (p.T != 4 || p.H != 2 * p.HO || p.W != 2 * p.WO || p.proq != S2K_PRO_NONE || p.gateq) {
            set_error("wgrad: gather geometry"); return S2K_EINVAL;
        }
        p.NP = 32;
        p.PSTR = p.NP + 1;
        p.CSQ = 4 * p.NP + 1;
        p.ntiles = (int)cdiv64(npix, p.NP);
        if (p.M <= 32 || p.C <= 32) return launch_wg<WG_GATHER, 4, 1, 1, 2, 1, 2, 32, 1>(p, st);
        return launch_wg<WG_GATHER, 4, 1, 2, 2, 2, 1, 32, 1>(p, st);
    }
    if (p.T == 1 && p.S == 1) {
        if (p.H != p.HO || p.W != p.WO) { set_error("wgrad: 1x1 geometry"); return S2K_EINVAL; }
        p.NP = 64;
        p.PSTR = p.NP + 1;
        p.CSQ = p.NP + 1;
        p.ntiles = (int)cdiv64(npix, p.NP);
        if (p.M <= 32 && p.C <= 32) {
            // thin layers on large maps (the classifier, 24-channel blocks at 128x128): 256 pixels per stage, 4x fewer barriers per byte
            static const int thin_np = tune_int("S2K_WG_THIN_NP", 256);
            if (thin_np == 256 && npix >= 262144) {
                p.NP = 256; p.PSTR = p.NP + 1; p.CSQ = p.NP + 1; p.ntiles = (int)cdiv64(npix, p.NP);
                return launch_wg<WG_PIX, 1, 1, 1, 1, 1, 4, 256, 1>(p, st);
            }
            return launch_wg<WG_PIX, 1, 1, 1, 1, 1, 4, 64, 1>(p, st);
        }
        // tile edge per side: 128 unless it pads the side by more than 12 % (176 -> 256 wastes 45 %, 3 x 64 = 192 wastes 9 %)
        auto edge = [](int n) { return (n > 64 && (double)cdiv(n, 128) * 128 / n <= 1.12) ? 128 : 64; };
        const int em = edge(p.M), ec = edge(p.C);
        if (em == 128 && ec == 128) return launch_wg<WG_PIX, 1, 2, 2, 2, 2, 1, 64, 1>(p, st);
        if (em == 128) return launch_wg<WG_PIX, 1, 2, 1, 2, 2, 1, 64, 1>(p, st);
        if (ec == 128) return launch_wg<WG_PIX, 1, 1, 2, 2, 2, 1, 64, 1>(p, st);
        return launch_wg<WG_PIX, 1, 1, 1, 2, 2, 1, 64, 1>(p, st);
    }
    if (p.T != 9) { set_error("wgrad: only 1x1, 3x3 and 2x2-transpose kernels are on this path"); return S2K_EINVAL; }
    // 3x3 (stride 1 pad 1, or the stride-2 TF-SAME stem): rectangular pixel tiles; thin layers (few
    // channels, huge maps) take 128-pixel tiles, the rest 64 (two workgroups per CU by LDS)
    static const int wide = tune_int("S2K_WG_WIDE", 0);
    const bool thin = (p.M <= 32) || (p.M <= 64 && p.C <= 32);   // few channels on one side: 128-pixel tiles, waves split the pixels
    const int NPX = (thin || wide) ? 128 : 64;
    int XW = p.WO <= NPX ? p.WO : NPX;
    int R = NPX / ((XW + 1) & ~1);
    if (R > p.HO) R = p.HO;
    if (R < 1) R = 1;
    for (;; --R) {
        p.R = R; p.XW = XW; p.XWe = (XW + 1) & ~1;
        p.IR = (R - 1) * p.S + p.KH;
        p.IC = (p.XWe - 1) * p.S + p.KW;
        p.WS = p.IC;
        if (p.IR * p.WS <= NTHREADS * WG_EPT_MAX) break;
        if (R == 1) {
            if (XW > 16) { XW /= 2; R = 2; continue; }
            set_error("wgrad: halo tile does not fit"); return S2K_EINVAL;
        }
    }
    p.CSQ = (p.IR * p.WS) | 1;
    p.PSTR = (p.R * p.XWe) | 1;
    p.tiles_x = cdiv(p.WO, p.XW);
    p.tiles_y = cdiv(p.HO, p.R);
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.NP = 0;
    if (thin && p.M <= 32 && p.C <= 32) return launch_wg<WG_SPATIAL, 9, 1, 1, 1, 1, 4, 128, 2>(p, st);
    if (thin && p.M <= 32) return launch_wg<WG_SPATIAL, 9, 1, 1, 1, 2, 2, 128, 2>(p, st);
    if (thin) return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 1, 2, 128, 2>(p, st);   // 64 x 32: the stem (48 x 13) and 64 x 24 skips no longer pad C to 64
    if (wide) return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 128, 2>(p, st);
    if (p.IR * p.WS <= NTHREADS && p.WS == 66 && p.KW == 3) return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 64, 1, 66>(p, st);   // 64-wide tiles
    if (p.IR * p.WS <= NTHREADS) return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 64, 1>(p, st);
    return launch_wg<WG_SPATIAL, 9, 1, 1, 2, 2, 1, 64, 2>(p, st);
}

}  // namespace s2k
