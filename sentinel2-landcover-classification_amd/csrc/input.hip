// GPU-side input pipeline: raw int16 Sentinel-2 tiles + uint8 label rasters -> the network's
// fp32 NCHW crops and int64 label maps, one pass.  Replaces, per sample, the reference's CPU chain
//   s2osm_dataset.py:51-71   cnes_transform (np.vectorize remap, cnes_labell_mappings.py:85-95), channel-last round trip,
//                            .float(), .long()
//   s2osm_datamodule.py:75-87  A.RandomCrop / A.CenterCrop -> A.HorizontalFlip -> A.VerticalFlip -> A.Normalize
// which runs on one loader worker in the reference.  HBM-bound: 2 B read + 4 B written per image element; lanes run along
// the output row (128-B coalesced stores; the int16 reads of a row are contiguous too, reversed under a horizontal flip).
// The crop offsets and flip decisions come from the host (PARAMS), like every other random draw of this library.
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

struct PrepP {
    const short* raw;
    const unsigned char* labels;
    const int* params;     // [B][4] = {src, y0, x0, flips}
    const float* norm;     // [2][C] = {mean * max_pixel_value, 1 / (std * max_pixel_value)}
    const int* lut;        // [256]
    float* x;
    long long* y;
    int B, C, H, W, S, NSRC;
};

// grid: (row groups, C + 1 planes, B); plane C is the label map.  One thread per 4 output columns.
__global__ void __launch_bounds__(NTHREADS) tile_prep_kernel(const PrepP p) {
    const int b = blockIdx.z, plane = blockIdx.y;
    const int q = p.S >> 2;                                   // column quads per row (S % 4 == 0, host check)
    const int e = blockIdx.x * NTHREADS + threadIdx.x;
    if (e >= p.S * q) return;
    const int i = e / q, j = (e - i * q) * 4;
    const int src = p.params[4 * b], y0 = p.params[4 * b + 1], x0 = p.params[4 * b + 2], flips = p.params[4 * b + 3];
    const int si = y0 + ((flips & 2) ? p.S - 1 - i : i);
    const int sj = x0 + ((flips & 1) ? p.S - 1 - j : j);      // source column of output column j; j + k maps to sj -/+ k
    const int step = (flips & 1) ? -1 : 1;
    if (plane < p.C) {
        const short* row = p.raw + (((int64_t)src * p.C + plane) * p.H + si) * p.W;
        const float m = p.norm[plane], d = p.norm[p.C + plane];
        float4 o;
        o.x = __fmul_rn(__fsub_rn((float)row[sj], m), d);
        o.y = __fmul_rn(__fsub_rn((float)row[sj + step], m), d);
        o.z = __fmul_rn(__fsub_rn((float)row[sj + 2 * step], m), d);
        o.w = __fmul_rn(__fsub_rn((float)row[sj + 3 * step], m), d);
        *reinterpret_cast<float4*>(p.x + (((int64_t)b * p.C + plane) * p.S + i) * p.S + j) = o;
    } else if (p.y) {
        const unsigned char* row = p.labels + ((int64_t)src * p.H + si) * p.W;
        long long* dst = p.y + ((int64_t)b * p.S + i) * p.S + j;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = p.lut[row[sj + k * step]];
    }
}

// PARAMS are not range-checked on the device (that would cost a sync or a branch per element): the host mirror
// (data/gpu_pipeline.py) draws / validates them before upload: 0 <= src < NSRC, 0 <= y0 <= H - S, 0 <= x0 <= W - S.
int launch_tile_prep(const S2kOp& op, const Ctx& c) {
    PrepP p{};
    p.raw = ref_ptr<const short>(c, op.t[S2K_TILE_PREP_T_RAW]);
    p.labels = ref_ptr<const unsigned char>(c, op.t[S2K_TILE_PREP_T_LABELS]);
    p.params = ref_ptr<const int>(c, op.t[S2K_TILE_PREP_T_PARAMS]);
    p.norm = ref_ptr<const float>(c, op.t[S2K_TILE_PREP_T_NORM]);
    p.lut = ref_ptr<const int>(c, op.t[S2K_TILE_PREP_T_LUT]);
    p.x = ref_ptr<float>(c, op.t[S2K_TILE_PREP_T_X]);
    p.y = ref_ptr<long long>(c, op.t[S2K_TILE_PREP_T_Y]);
    const void* ptrs[] = {p.raw, p.labels, p.params, p.norm, p.lut, p.x, p.y};
    for (const void* q : ptrs)
        if (q == reinterpret_cast<const void*>(1)) { set_error("tile_prep: tensor references a null base"); return S2K_EFAULT; }
    p.B = op.d[S2K_TILE_PREP_D_B]; p.C = op.d[S2K_TILE_PREP_D_C]; p.H = op.d[S2K_TILE_PREP_D_H]; p.W = op.d[S2K_TILE_PREP_D_W];
    p.S = op.d[S2K_TILE_PREP_D_S]; p.NSRC = op.d[S2K_TILE_PREP_D_NSRC];
    if (!p.raw || !p.params || !p.norm || !p.x || (p.y && (!p.labels || !p.lut))) { set_error("tile_prep: missing tensor"); return S2K_EINVAL; }
    if (p.B <= 0 || p.C <= 0 || p.S <= 0 || (p.S & 3) || p.S > p.H || p.S > p.W || p.NSRC <= 0 || p.B > 65535 || p.C >= 65535) {
        set_error("tile_prep: bad dims (crop size must be a multiple of 4 and fit the tile)"); return S2K_EINVAL;
    }
    const int per_plane = p.S * (p.S >> 2);
    hipLaunchKernelGGL(tile_prep_kernel, dim3(cdiv(per_plane, NTHREADS), p.C + (p.y ? 1 : 0), p.B), dim3(NTHREADS), 0, c.stream, p);
    return S2K_OK;
}

}  // namespace s2k
