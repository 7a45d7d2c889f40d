// Weight gradients of the dense convs on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, f32 accumulate): the
// bf16-MIXED mode (see conv_bf16.hip; reported separately from the f32 parity path).  Same stage record and arithmetic as
// wgrad.hip / wgrad_pc.hip -  dW[tap][m][c] += sum_pix Ppro[m][pix] * Qpro[c][pix + tap]  - with the two MFMA operands rounded to
// bf16 when a tile is written to LDS; partial sums of the pixel splits are added with f32 atomics as before.  Replaces the same
// reference code (ATen convolution_backward grad_weight of efficientnet_unet.py:168-176,319-372 under autocast).
//
// The contraction runs over PIXELS, and both operands are pixel-contiguous in memory ([row][pixels], f32): a lane's MFMA operand
// - 8 consecutive pixels of its row - is 32 contiguous bytes of global memory.  A staging thread loads them (two 16-byte
// loads), applies the prologue (BatchNorm affine + activation (+ SE gate), per-row constants), rounds to bf16 and stores ONE
// 16-byte unit [pixel octet][row]; lanes are laid out 8 rows x 8 octets per wave instruction, so a wave reads 8 rows x 256
// contiguous bytes and the 8 lanes of a ds_write_b128 group write 8 consecutive units (conflict-free).
//
// 3x3 (stride 1, pad 1): the nine taps read the SAME aligned pixel octets of the three halo rows, shifted by -1 / 0 / +1
// pixel.  A shift by one pixel is 16 bits inside the 4-dword fragment: v_alignbit_b32 between neighbouring dwords, with the
// last dword of the octet to the left / the first of the octet to the right at the ends (the tile's halo columns are one-pixel
// units).  Five v_alignbit per halo row and k-step give all three horizontal taps.
//
// With bf16 MFMAs a weight gradient is bound by HBM / L2 bandwidth (32 flop per byte at 128 x 128 tiles against 312 flop per
// byte of the chip), so the kernel is built for bytes in flight: 4 waves per workgroup, no role split, 2-3 workgroups per CU.
#include <algorithm>

#include "common.h"
#include "wgrad.h"

namespace s2k {

typedef __bf16 wb16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb16x2 __attribute__((ext_vector_type(2)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wpk(float lo, float hi) {
    wf32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wb16x2));
}
__device__ __forceinline__ uint32_t shr16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }   // {hi, lo} >> 16

// prologue of 8 consecutive pixels of one row -> one bf16 unit; `valid` = the pixels exist (zero padding / ragged tiles: the
// reference pads ACTIVATED maps with zeros)
__device__ __forceinline__ wu32x4 pro_unit(const f32x4& a, const f32x4& b, int pro, float sc, float sh, float gate, bool valid) {
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (pro != S2K_PRO_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = apply_pro(v[i], pro, sc, sh);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= gate;
    wu32x4 w = {wpk(v[0], v[1]), wpk(v[2], v[3]), wpk(v[4], v[5]), wpk(v[6], v[7])};
    if (!valid) w = wu32x4{0u, 0u, 0u, 0u};
    return w;
}

// The four waves are arranged WVM (m) x WVC (c) x WVK (k): each computes WM x WN accumulator tiles of 32 x 32 per tap over the
// k-steps s = wk, wk + WVK, ... of a pixel tile.  WVK > 1 (thin layers: a 32-channel side leaves one 32 x 32 tile per tap) splits the
// pixels of a tile over waves; every wave adds its partial sums in the combine.
// Pixel tile: 1x1: XW consecutive pixels (R = 1); 3x3: R rows x XW pixels.
// 1x1: two workgroups per CU; 3x3 (144 accumulator registers per lane + the next tile's staging registers): one.
// P16 (no prologue on P): P is stored as bf16 [B][M][HW] (a BN_BWD_APPLY with OUT_BF16 wrote it): a lane's 8 pixels are ONE
// 16-byte load and ARE the LDS unit - the values the f32 path would round to.
template <int MODE, int WVM, int WVC, int WM, int WN, int R, int XW, bool P16 = false>
__global__ void __launch_bounds__(256, MODE == WG_PIX ? 2 : 1) wgrad_bf16_kernel(const WgradP p) {
    constexpr bool PIX = MODE == WG_PIX;

    constexpr int NT = 256;
    constexpr int T = PIX ? 1 : 9;
    constexpr int WVK = 4 / (WVM * WVC);
    constexpr int BM = WVM * WM * 32, BC = WVC * WN * 32;
    constexpr int NPJ = PIX ? XW : R * XW;           // pixels per tile
    constexpr int XO = XW / 8;                       // pixel octets per tile row
    constexpr int NPO = PIX ? XO : R * XO;           // P octets per row m
    constexpr int QR = PIX ? 1 : R + 2;              // Q rows of the tile (halo rows)
    constexpr int QXO = PIX ? XO : XO + 2;           // Q unit slots per row: left halo pixel, XO octets, right halo pixel
    constexpr int NQO = PIX ? XO : QR * XO;          // Q full octets per channel
    static_assert(WVM * WVC * WVK == 4 && (NPO / 2) % WVK == 0, "wave grid");
    constexpr int P_UNITS = NPO * BM;
    constexpr int PRG = BM / 32, QRG = BC / 32;      // row groups (8 rows) per wave: wave w owns groups w, w + 4, ...  (BM, BC >= 32)
    constexpr int POB = (NPO + 7) / 8, QOB = (NQO + 7) / 8;      // octet blocks (8 octets = one wave instruction's width)
    constexpr int NHI = PIX ? 0 : (BC * QR * 2 + NT - 1) / NT;   // halo-pixel items per thread
    static_assert(PIX || (XW % 8 == 0 && NPO % 2 == 0), "3x3 tile: whole octets, whole k-steps");
    extern __shared__ __attribute__((aligned(16))) wu32x4 smem_w[];
    wu32x4* Ps = smem_w;
    wu32x4* Qs = smem_w + P_UNITS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int mc = p.n_mtiles * p.n_ctiles;
    const int v = wg_xcd_remap(blockIdx.x, gridDim.x);
    const int split = v / mc, tl = v - split * mc;
    const int mt = tl % p.n_mtiles, ct = tl / p.n_mtiles;
    const int m0 = mt * BM, c0 = ct * BC;
    const int tile_begin = split * p.tiles_per_split;
    int tile_end = tile_begin + p.tiles_per_split;
    if (tile_end > p.ntiles) tile_end = p.ntiles;

    // ---- per-thread staging rows: fixed for the whole kernel (a wave instruction = 8 rows x 8 octets) -------------------------
    const int r8 = lane & 7, o8 = lane >> 3;
    int prow[PRG], qrow[QRG];
    float psc[PRG], psh[PRG], qsc[QRG], qsh[QRG];
#pragma unroll
    for (int j = 0; j < PRG; ++j) {
        prow[j] = min(m0 + (wave + 4 * j) * 8 + r8, p.M - 1);        // rows past M re-read the last one (discarded at the combine)
        psc[j] = p.prop != S2K_PRO_NONE ? p.bnvp[prow[j]] : 1.0f;
        psh[j] = p.prop != S2K_PRO_NONE ? p.bnvp[p.M + prow[j]] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < QRG; ++j) {
        qrow[j] = min(c0 + (wave + 4 * j) * 8 + r8, p.C - 1);
        qsc[j] = p.proq != S2K_PRO_NONE ? p.bnvq[qrow[j]] : 1.0f;
        qsh[j] = p.proq != S2K_PRO_NONE ? p.bnvq[p.C + qrow[j]] : 0.0f;
    }
    const int64_t ntot = (int64_t)p.B * p.HWp;
    const bool img_local = !PIX || (p.HWp % NPJ) == 0;
    const uint32_t p_rstep = (uint32_t)p.HWp * (P16 ? 2u : 4u), q_rstep = (uint32_t)p.HWq * 4u;

    f32x16 acc[T][WM][WN];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.0f;
    const int wk = wave % WVK, wmn = wave / WVK;
    const int wm0 = (wmn / WVC) * (WM * 32), wc0 = (wmn % WVC) * (WN * 32);

    // ---- staging registers: the WHOLE next tile is fetched while the current one is multiplied ---------------------------------
    f32x4 pxa[POB][PRG], pxb[POB][PRG], qxa[QOB][QRG], qxb[QOB][QRG];
    float qgate[QOB][QRG];
    bool pok[POB], qok[QOB];
    float hval[NHI > 0 ? NHI : 1];

    auto fetch = [&](int tile) {
        int tb = 0, y0 = 0, x0 = 0;
        if (PIX) {
            tb = (int)(((int64_t)tile * NPJ) / p.HWp);      // first image the tile touches: offsets span the tile's images only
        } else {
            const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y;
            tb = tile / (p.tiles_x * p.tiles_y);
            y0 = ty * R; x0 = tx * XW;
        }
        const rsrc_t rp = P16 ? make_rsrc(reinterpret_cast<const uint16_t*>(p.p) + (int64_t)tb * p.M * p.HWp, (int64_t)(img_local ? 1 : p.B - tb) * p.M * p.HWp * 2)
                              : make_rsrc(p.p + (int64_t)tb * p.M * p.HWp, (int64_t)(img_local ? 1 : p.B - tb) * p.M * p.HWp * 4);
        const rsrc_t rq = make_rsrc(p.q + (int64_t)tb * p.C * p.HWq, (int64_t)(img_local ? 1 : p.B - tb) * p.C * p.HWq * 4);
#pragma unroll
        for (int ob = 0; ob < POB; ++ob) {
            // byte offset of this lane's pixel octet of the P tile inside a row (image-relative), or "no such pixels"
            const int oct = ob * 8 + o8;
            uint32_t off = 0;
            bool ok = oct < NPO;
            if (PIX) {
                const int64_t n = (int64_t)tile * NPJ + 8 * oct;
                ok = ok && n < ntot;
                const int64_t nn = ok ? n : 0;
                const int b = (int)(nn / p.HWp);
                off = (uint32_t)((int64_t)(b - tb) * p.M * p.HWp + (nn - (int64_t)b * p.HWp)) * (P16 ? 2u : 4u);
            } else {
                const int r = oct / XO, k = oct % XO;
                off = (uint32_t)((y0 + r) * p.WO + x0 + 8 * k) * (P16 ? 2u : 4u);
                ok = ok && y0 + r < p.HO;
            }
            pok[ob] = ok;
#pragma unroll
            for (int j = 0; j < PRG; ++j) {
                const uint32_t a = ok ? off + (uint32_t)prow[j] * p_rstep : BUF_OOB;
                pxa[ob][j] = bload4(rp, a);                                   // (P16: the whole unit, eight bf16)
                if constexpr (!P16) pxb[ob][j] = bload4(rp, ok ? a + 16u : BUF_OOB);
            }
        }
#pragma unroll
        for (int ob = 0; ob < QOB; ++ob) {
            const int oct = ob * 8 + o8;
            uint32_t off = 0;
            bool ok = oct < NQO;
            int b = tb;
            if (PIX) {
                const int64_t n = (int64_t)tile * NPJ + 8 * oct;
                ok = ok && n < ntot;
                const int64_t nn = ok ? n : 0;
                b = (int)(nn / p.HWq);
                off = (uint32_t)((int64_t)(b - tb) * p.C * p.HWq + (nn - (int64_t)b * p.HWq)) * 4u;
            } else {
                const int hr = oct / XO, k = oct % XO;
                const int iy = y0 - 1 + hr;
                off = (uint32_t)(iy * p.W + x0 + 8 * k) * 4u;
                ok = ok && iy >= 0 && iy < p.H;
            }
            qok[ob] = ok;
#pragma unroll
            for (int j = 0; j < QRG; ++j) {
                const uint32_t a = ok ? off + (uint32_t)qrow[j] * q_rstep : BUF_OOB;
                qxa[ob][j] = bload4(rq, a);
                qxb[ob][j] = bload4(rq, ok ? a + 16u : BUF_OOB);
                qgate[ob][j] = (PIX && p.gateq && ok) ? p.gateq[(int64_t)b * p.C + qrow[j]] : 1.0f;
            }
        }
        if constexpr (!PIX) {
            // the tile's halo columns: one pixel left of octet 0 and one right of the last octet, per (channel, halo row)
#pragma unroll
            for (int i = 0; i < NHI; ++i) {
                const int idx = tid + NT * i;
                const int c = idx % BC, rest = idx / BC;
                const int side = rest & 1, hr = rest >> 1;
                const int iy = y0 - 1 + hr, ix = side ? x0 + XW : x0 - 1;
                const bool ok = idx < BC * QR * 2 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const int gc = min(c0 + c, p.C - 1);
                float x = bload(rq, ok ? (uint32_t)(iy * p.W + ix) * 4u + (uint32_t)gc * q_rstep : BUF_OOB);
                if (p.proq != S2K_PRO_NONE) x = apply_pro(x, p.proq, p.bnvq[gc], p.bnvq[p.C + gc]);
                hval[NHI > 0 ? i : 0] = ok ? x : 0.0f;
            }
        }
    };
    // prologue + bf16 rounding + LDS: nothing here depends on the tile (validity travels in pok / qok)
    auto commit = [&]() {
#pragma unroll
        for (int ob = 0; ob < POB; ++ob) {
            const int oct = ob * 8 + o8;
            if (oct < NPO)
#pragma unroll
                for (int j = 0; j < PRG; ++j) {
                    if constexpr (P16) Ps[oct * BM + (wave + 4 * j) * 8 + r8] = pok[ob] ? __builtin_bit_cast(wu32x4, pxa[ob][j]) : wu32x4{0u, 0u, 0u, 0u};
                    else Ps[oct * BM + (wave + 4 * j) * 8 + r8] = pro_unit(pxa[ob][j], pxb[ob][j], p.prop, psc[j], psh[j], 1.0f, pok[ob]);
                }
        }
#pragma unroll
        for (int ob = 0; ob < QOB; ++ob) {
            const int oct = ob * 8 + o8;
            if (oct < NQO) {
                const int slot = PIX ? oct : (oct / XO) * QXO + 1 + (oct % XO);
#pragma unroll
                for (int j = 0; j < QRG; ++j)
                    Qs[slot * BC + (wave + 4 * j) * 8 + r8] = pro_unit(qxa[ob][j], qxb[ob][j], p.proq, qsc[j], qsh[j], qgate[ob][j], qok[ob]);
            }
        }
        if constexpr (!PIX) {
#pragma unroll
            for (int i = 0; i < NHI; ++i) {
                const int idx = tid + NT * i;
                if (idx < BC * QR * 2) {
                    const int c = idx % BC, rest = idx / BC;
                    const int side = rest & 1, hr = rest >> 1;
                    const float x = hval[NHI > 0 ? i : 0];
                    // left halo pixel = element 7 of the unit left of octet 0; right halo pixel = element 0 of the unit after the last
                    Qs[(hr * QXO + (side ? XO + 1 : 0)) * BC + c] = side ? wu32x4{wpk(x, 0.0f), 0u, 0u, 0u} : wu32x4{0u, 0u, 0u, wpk(0.0f, x)};
                }
            }
        }
    };

    if (tile_begin < tile_end) fetch(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        commit();
        __syncthreads();
        if (tile + 1 < tile_end) fetch(tile + 1);                   // in flight while this tile is multiplied

        // ================================================ multiply =============================================================
#pragma unroll
        for (int s0 = 0; s0 < NPO / 2; s0 += WVK) {
            const int s = s0 + wk;                                      // this wave's k-step (16 pixels)
            const int o = 2 * s + lh;                                   // this lane half's pixel octet
            wb16x8 a[WM];
#pragma unroll
            for (int rm = 0; rm < WM; ++rm) a[rm] = __builtin_bit_cast(wb16x8, Ps[o * BM + wm0 + rm * 32 + l31]);
            if constexpr (PIX) {
#pragma unroll
                for (int rn = 0; rn < WN; ++rn) {
                    const wb16x8 b = __builtin_bit_cast(wb16x8, Qs[o * BC + wc0 + rn * 32 + l31]);
#pragma unroll
                    for (int rm = 0; rm < WM; ++rm)
                        acc[0][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm], b, acc[0][rm][rn], 0, 0, 0);
                }
            } else {
                const int r = o / XO, k = o % XO;
#pragma unroll
                for (int rn = 0; rn < WN; ++rn)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const wu32x4* base = Qs + ((r + dy) * QXO + 1 + k) * BC + wc0 + rn * 32 + l31;
                        const wu32x4 cur = base[0];
                        const uint32_t pl = reinterpret_cast<const uint32_t*>(base - BC)[3];     // last dword of the unit to the left
                        const uint32_t nf = reinterpret_cast<const uint32_t*>(base + BC)[0];     // first dword of the unit to the right
                        const uint32_t s01 = shr16(cur[1], cur[0]), s12 = shr16(cur[2], cur[1]), s23 = shr16(cur[3], cur[2]);
                        const wu32x4 left = {shr16(cur[0], pl), s01, s12, s23};                   // pixels x - 1 .. x + 6
                        const wu32x4 right = {s01, s12, s23, shr16(nf, cur[3])};                  // pixels x + 1 .. x + 8
                        const wb16x8 b0 = __builtin_bit_cast(wb16x8, left), b1 = __builtin_bit_cast(wb16x8, cur), b2 = __builtin_bit_cast(wb16x8, right);
#pragma unroll
                        for (int rm = 0; rm < WM; ++rm) {
                            acc[dy * 3 + 0][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm], b0, acc[dy * 3 + 0][rm][rn], 0, 0, 0);
                            acc[dy * 3 + 1][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm], b1, acc[dy * 3 + 1][rm][rn], 0, 0, 0);
                            acc[dy * 3 + 2][rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm], b2, acc[dy * 3 + 2][rm][rn], 0, 0, 0);
                        }
                    }
            }
        }
        __syncthreads();
    }

    // ---------------- combine: wgs[t][m][c] += acc (k-waves reduced through LDS first: wgrad.h) -----------------------------------
    if (p.exp & 2) return;          // tuning builds only (S2K_WG_EXP)
    if (tile_begin >= tile_end) return;
    wg_combine<T, WM, WN, WVK>(p, acc, reinterpret_cast<float*>(smem_w), wk, wmn, lane, m0, c0, wm0, wc0);
}

// -------------------------------------------------------------------------------------------------
template <int MODE, int WVM, int WVC, int WM, int WN, int R, int XW, bool P16 = false>
static int launch_wb16(WgradP& p, hipStream_t st) {
    constexpr bool PIX = MODE == WG_PIX;
    constexpr int BM = WVM * WM * 32, BC = WVC * WN * 32;
    constexpr int NPJ = PIX ? XW : R * XW;
    constexpr int XO = XW / 8, NPO = PIX ? XO : R * XO, QR = PIX ? 1 : R + 2, QXO = PIX ? XO : XO + 2;
    constexpr size_t lds = (size_t)(NPO * BM + QR * QXO * BC) * 16;
    static_assert(lds <= 160 * 1024, "LDS image");
    p.n_mtiles = cdiv(p.M, BM);
    p.n_ctiles = cdiv(p.C, BC);
    if (PIX) {
        p.NP = NPJ;
        p.ntiles = (int)cdiv64((int64_t)p.B * p.HWp, NPJ);
    } else {
        p.R = R; p.XW = XW; p.XWe = XW;
        p.tiles_x = p.WO / XW;
        p.tiles_y = cdiv(p.HO, R);
        p.ntiles = p.B * p.tiles_x * p.tiles_y;
        p.NP = 0;
    }
    {   // 32-bit buffer offsets: one image below 2 GiB when tiles are image-local, else the whole tensor
        const int64_t span = (!PIX || (p.HWp % NPJ) == 0) ? 1 : std::min<int64_t>(p.B, (NPJ - 2) / p.HWp + 2);   // images one pixel tile touches
        const int64_t need = std::max((int64_t)p.M * p.HWp, (int64_t)p.C * p.HWq) * 4 * span;
        if (need >= 0x7ffffff0ll) return 1;
    }
    auto kern = wgrad_bf16_kernel<MODE, WVM, WVC, WM, WN, R, XW, P16>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // Pixel splits.  Every split ends in an atomic combine of its accumulator tiles, and float atomics run at ~1.3 TB/s chip-wide
    // (MI355X_MICROARCH.md): with bf16 MFMAs the combine of a 512 x 512 x 9 gradient split 12 ways (113 MB of adds, 87 us) cost
    // more than the whole contraction.  So: just enough splits to give every CU its workgroups (one per CU for the 3x3 kernels,
    // two for 1x1), whole rounds of them, and at least 4 pixel tiles per split.
    const int mc = p.n_mtiles * p.n_ctiles;
    static const int slots_exp = tune_int("S2K_WB16_SLOTS", 0);
    const int slots = slots_exp > 0 ? slots_exp : (PIX ? 512 : 256);
    int splits = std::max(1, slots / mc);
    splits = std::min(splits, std::max(1, p.ntiles / 4));
    if (splits > 65535) splits = 65535;
    p.tiles_per_split = cdiv(p.ntiles, splits);
    splits = cdiv(p.ntiles, p.tiles_per_split);
    hipLaunchKernelGGL(kern, dim3(mc * splits), dim3(256), lds, st, p);
    g_s2k_variant = 2;
    return S2K_OK;
}

// S2K_OK = launched, 1 = not one of its shapes (the caller takes the f32 kernels), < 0 = error
int launch_wgrad_bf16(WgradP& p, int mode, hipStream_t st) {
    if (mode != S2K_MODE_CONV || p.S != 1 || p.H != p.HO || p.W != p.WO || p.gatep) return 1;
    auto pro_ok = [](int pro) { return pro == S2K_PRO_NONE || pro == S2K_PRO_RELU || pro == S2K_PRO_SILU || pro == S2K_PRO_AFFINE || pro == S2K_PRO_GELU; };
    if (!pro_ok(p.prop) || !pro_ok(p.proq)) return 1;
    // tile edge per side: 32 for thin layers, 128 (1x1 only) unless it pads the side by more than 12 %, else 64
    auto edge = [](int n) { return n <= 32 ? 32 : ((n > 64 && (double)cdiv(n, 128) * 128 / n <= 1.12) ? 128 : 64); };
    const int em = edge(p.M), ec = edge(p.C);
    if (p.T == 1) {
        if ((p.HWp & 7) || (int64_t)p.B * p.HWp < 512) return 1;
        // thin sides: the waves split the pixels of 256-pixel tiles instead of the (single) accumulator tile
        if (p.p_bf16) {       // P stored as bf16 (opdefs WGRAD.P_BF16)
            if (p.prop != S2K_PRO_NONE) { set_error("wgrad: P_BF16 with a prologue on P"); return S2K_EINVAL; }
            if (em == 32 && ec == 32) return launch_wb16<WG_PIX, 1, 1, 1, 1, 1, 256, true>(p, st);
            if (em == 32) return launch_wb16<WG_PIX, 1, 2, 1, 1, 1, 128, true>(p, st);
            if (ec == 32) return launch_wb16<WG_PIX, 2, 1, 1, 1, 1, 128, true>(p, st);
            if (em == 128 && ec == 128) return launch_wb16<WG_PIX, 2, 2, 2, 2, 1, 64, true>(p, st);
            if (em == 128) return launch_wb16<WG_PIX, 2, 2, 2, 1, 1, 64, true>(p, st);
            if (ec == 128) return launch_wb16<WG_PIX, 2, 2, 1, 2, 1, 64, true>(p, st);
            return launch_wb16<WG_PIX, 2, 2, 1, 1, 1, 64, true>(p, st);
        }
        if (em == 32 && ec == 32) return launch_wb16<WG_PIX, 1, 1, 1, 1, 1, 256>(p, st);
        if (em == 32) return launch_wb16<WG_PIX, 1, 2, 1, 1, 1, 128>(p, st);          // 32 x 64, two waves per k-step
        if (ec == 32) return launch_wb16<WG_PIX, 2, 1, 1, 1, 1, 128>(p, st);          // 64 x 32
        if (em == 128 && ec == 128) return launch_wb16<WG_PIX, 2, 2, 2, 2, 1, 64>(p, st);
        if (em == 128) return launch_wb16<WG_PIX, 2, 2, 2, 1, 1, 64>(p, st);
        if (ec == 128) return launch_wb16<WG_PIX, 2, 2, 1, 2, 1, 64>(p, st);
        return launch_wb16<WG_PIX, 2, 2, 1, 1, 1, 64>(p, st);
    }
    if (p.T != 9 || p.KH != 3 || p.KW != 3 || p.PT != 1 || p.PL != 1 || p.gateq || p.prop != S2K_PRO_NONE) return 1;
#define WB16_3X3(...) (p.p_bf16 ? launch_wb16<WG_SPATIAL, __VA_ARGS__, true>(p, st) : launch_wb16<WG_SPATIAL, __VA_ARGS__>(p, st))
    if (p.WO % 64 == 0) {
        if (em == 32 && ec == 32) return WB16_3X3(1, 1, 1, 1, 4, 64);     // 32 x 32 tile, 4 x 64 pixels, waves split the pixels
        if (em == 32) return WB16_3X3(1, 2, 1, 1, 4, 64);
        if (ec == 32) return WB16_3X3(2, 1, 1, 1, 4, 64);
        return WB16_3X3(2, 2, 1, 1, 4, 64);      // 4 x 64 pixel tiles: halo 6 rows per 4
    }
    if (p.WO % 56 == 0) return WB16_3X3(2, 2, 1, 1, 2, 56);      // 224-pixel inputs: 56 / 112 / 224 wide maps
    if (p.WO == 32) return WB16_3X3(2, 2, 1, 1, 4, 32);
    if (p.WO == 16) return WB16_3X3(2, 2, 1, 1, 8, 16);
#undef WB16_3X3
    return 1;
}

}  // namespace s2k
