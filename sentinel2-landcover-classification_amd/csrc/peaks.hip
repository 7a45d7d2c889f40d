// Measured ceilings of THIS device, for bench.py's roofline line ("report roofline fractions against both
// spec and measured peaks"): the sustained rate of v_mfma_f32_32x32x2_f32 with every SIMD issuing back to back (operands
// in registers, 8 independent accumulators, 1 or 2 waves per SIMD), the shader clock the chip holds while doing so
// (s_memtime / s_memrealtime), and a float4 stream copy.  Not part of the hot path.
#include <algorithm>

#include "common.h"

namespace s2k {

__global__ void __launch_bounds__(256) mfma_peak_kernel(float* sink, unsigned long long* stamps, int iters) {
    f32x16 acc[8];
    const float a = 1.0f + 0.001f * (float)(threadIdx.x & 63), b = 0.5f - 0.002f * (float)(threadIdx.x & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;                       // keeps the accumulators alive; never true
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;                   // shader-clock cycles
        stamps[2 * blockIdx.x + 1] = r1 - r0;               // 100 MHz ticks
    }
}

// Stream copy: eight independent 16-byte loads in flight per lane before the first store (a loop of one dependent load / store
// pair per trip measured 4.6 TB/s; the hot path's own adam_kernel sustains 6.1 TB/s, which a "peak" must at least reach),
// every CU holding 32 waves.
__global__ void __launch_bounds__(256) copy_peak_kernel(const float4* src, float4* dst, int64_t n4) {
    // every workgroup streams 32 KB pieces (8 x 4 KB, contiguous) of the buffer, workgroups interleaved piece by piece
    const int64_t piece = 8 * 256;
    for (int64_t base = (int64_t)blockIdx.x * piece; base < n4; base += (int64_t)gridDim.x * piece) {
        const int64_t i = base + threadIdx.x;
        if (base + piece <= n4) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + i + k * 256));
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(dst + i + k * 256));
        } else {
            for (int64_t j = i; j < n4; j += 256) dst[j] = src[j];
        }
    }
}

}  // namespace s2k

using namespace s2k;

extern "C" int s2k_measure_peaks(void* scratch, size_t scratch_bytes, int waves_per_simd, double* mfma_tflops, double* mfma_clock_mhz,
                                 double* copy_gbps, void* stream) {
    if (!scratch || scratch_bytes < (64u << 20) || !mfma_tflops || !mfma_clock_mhz || !copy_gbps || waves_per_simd < 1 || waves_per_simd > 2) {
        set_error("measure_peaks: need >= 64 MiB of device scratch and 1 or 2 waves per SIMD");
        return S2K_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("measure_peaks: no device properties"); return S2K_EHIP; }
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;                 // 256 threads = one wave per SIMD of a CU
    float* sink = static_cast<float*>(scratch);
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 40000;                                 // 8 x 40000 MFMAs per wave: ~10 ms at full rate
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, sink, stamps, 2000);   // warm-up
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, sink, stamps, iters);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4.0 * 8.0 * iters * 4096.0;
    *mfma_tflops = flops / (ms * 1e-3) / 1e12;
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    *mfma_clock_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    // stream copy over half of the scratch each way (>= 32 MiB per direction; the caller sizes it past the caches)
    const int64_t n4 = (int64_t)((scratch_bytes - (1u << 20)) / 2 / 16);
    const float4* src = reinterpret_cast<const float4*>(static_cast<char*>(scratch) + (1u << 20));
    float4* dst = const_cast<float4*>(src) + n4;
    const int cblocks = (int)std::min<int64_t>(cdiv64(n4, 256 * 8), (int64_t)cus * 8);     // 32 waves per CU, >= 8 trips of 8 loads each
    hipLaunchKernelGGL(copy_peak_kernel, dim3(cblocks), dim3(256), 0, st, src, dst, n4);
    hipEventRecord(e0, st);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy_peak_kernel, dim3(cblocks), dim3(256), 0, st, src, dst, n4);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    *copy_gbps = 5.0 * 2.0 * (double)n4 * 16.0 / (ms * 1e-3) / 1e9;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("measure_peaks: %s", hipGetErrorString(e)); return S2K_EHIP; }
    return S2K_OK;
}

// ---- tuning builds only: what do LDS operand reads cost beside the f32 MFMA? -------------------------------------------
#ifdef S2K_TUNING
namespace s2k {
// 9 accumulators (the 3x3 weight-gradient shape); READS ds_read_b32 per 9 MFMAs feed the B operands (software-pipelined one
// group ahead, like the real kernels).  NW waves per SIMD all run the same stream.
template <int READS>
__global__ void __launch_bounds__(512) mfma_lds_kernel(float* sink, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const float* base = lds + (threadIdx.x & 31) * 65 + (threadIdx.x >> 5 & 1);
    float b0[9], b1[9];
    const float a = 1.0f + 0.001f * (float)(threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < 9; ++j) { b0[j] = base[j * 3]; b1[j] = base[j * 3 + 1]; }
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j < READS) b1[j] = base[(j * 7 + ((i * 2) & 63)) ];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j < READS) b0[j] = base[(j * 7 + ((i * 2 + 1) & 63)) ];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;
}
}  // namespace s2k

extern "C" int s2k_measure_mfma_lds(void* scratch, int reads, int waves_per_simd, double* tflops, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    const int blocks = prop.multiProcessorCount;
    const int threads = 256 * waves_per_simd;
    float* sink = static_cast<float*>(scratch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    auto launch = [&](int it) {
        if (reads == 0) hipLaunchKernelGGL(mfma_lds_kernel<0>, dim3(blocks), dim3(threads), 0, st, sink, it);
        else if (reads == 3) hipLaunchKernelGGL(mfma_lds_kernel<3>, dim3(blocks), dim3(threads), 0, st, sink, it);
        else if (reads == 5) hipLaunchKernelGGL(mfma_lds_kernel<5>, dim3(blocks), dim3(threads), 0, st, sink, it);
        else hipLaunchKernelGGL(mfma_lds_kernel<9>, dim3(blocks), dim3(threads), 0, st, sink, it);
    };
    launch(100);
    hipEventRecord(e0, st);
    launch(iters);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    *tflops = (double)blocks * 4.0 * waves_per_simd * 18.0 * iters * 4096.0 / (ms * 1e-3) / 1e12;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return S2K_OK;
}

// ---- what does one k-step of the conv kernels' MFMA loop cost (round 4)? ------------------------------------------------------------
// A wave owns WM x WN accumulator tiles; per k-step it needs WM A operands and WN B operands and issues WM * WN MFMAs.
// MODE 0: operands constant in registers; 1: operands produced by a VALU instruction per k-step; 2: operands from LDS, read DEPTH
// k-steps ahead (register sets rotate), interleaved with the MFMAs as in conv_pc_kernel / conv_dma_kernel.  One wave per SIMD.
namespace s2k {
template <int WM, int WN, int MODE, int DEPTH>
__global__ void __launch_bounds__(256) mfma_kstep_kernel(float* sink, unsigned long long* stamps, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int lane = threadIdx.x & 63;
    const float* Aa = lds + (lane >> 5) * 256 + (lane & 31);
    const float* Bb = lds + 8192 + (lane >> 5) * 128 + (lane & 31);
    float a[DEPTH + 1][WM], b[DEPTH + 1][WN];
#pragma unroll
    for (int d = 0; d <= DEPTH; ++d) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[d][i] = 1.0f + 0.001f * (float)(lane + i + d);
#pragma unroll
        for (int j = 0; j < WN; ++j) b[d][j] = 0.5f - 0.002f * (float)(lane + j + d);
    }
    auto fetch = [&](int s, float (&aa)[WM], float (&bb)[WN]) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < WM; ++i) aa[i] = aa[i] * 1.0001f;
#pragma unroll
            for (int j = 0; j < WN; ++j) bb[j] = bb[j] * 0.9999f;
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < WM; ++i) aa[i] = Aa[(s & 15) * 512 + i * 32];
#pragma unroll
            for (int j = 0; j < WN; ++j) bb[j] = Bb[(s & 15) * 256 + j * 32];
        }
    };
    auto mfmas = [&](const float (&aa)[WM], const float (&bb)[WN]) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[i], bb[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
    };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // DEPTH + 1 k-steps per trip; set d is consumed while the set DEPTH steps later is being fetched
#pragma unroll
        for (int d = 0; d <= DEPTH; ++d) {
            __builtin_amdgcn_sched_barrier(0);
            fetch(it * (DEPTH + 1) + d, a[(d + DEPTH) % (DEPTH + 1)], b[(d + DEPTH) % (DEPTH + 1)]);
            mfmas(a[d], b[d]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.678f) sink[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = c1 - c0;
}
}  // namespace s2k

// cycles per MFMA of the k-step loop (one wave per SIMD, every CU busy); wm x wn in {1x1, 2x1, 3x1, 2x2, 4x2}, mode 0..2, depth 1..3
extern "C" int s2k_measure_mfma_kstep(void* scratch, int wm, int wn, int mode, int depth, double* cycles_per_mfma, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    const int blocks = prop.multiProcessorCount;
    float* sink = static_cast<float*>(scratch);
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + 4096);
    const int iters = 2000;
    bool ok = false;
#define KSTEP(WMv, WNv, Mv, Dv) if (wm == WMv && wn == WNv && mode == Mv && depth == Dv) { ok = true; \
        hipLaunchKernelGGL((mfma_kstep_kernel<WMv, WNv, Mv, Dv>), dim3(blocks), dim3(256), 0, st, sink, stamps, 100); \
        hipLaunchKernelGGL((mfma_kstep_kernel<WMv, WNv, Mv, Dv>), dim3(blocks), dim3(256), 0, st, sink, stamps, iters); }
#define KSTEP_T(WMv, WNv) KSTEP(WMv, WNv, 0, 1) KSTEP(WMv, WNv, 1, 1) KSTEP(WMv, WNv, 2, 1) KSTEP(WMv, WNv, 2, 2) KSTEP(WMv, WNv, 2, 3)
    KSTEP_T(1, 1) KSTEP_T(2, 1) KSTEP_T(3, 1) KSTEP_T(2, 2) KSTEP_T(4, 2)
#undef KSTEP_T
#undef KSTEP
    if (!ok) { set_error("measure_mfma_kstep: configuration not instantiated"); return S2K_EINVAL; }
    if (hipStreamSynchronize(st) != hipSuccess) return S2K_EHIP;
    unsigned long long h = 0;
    (void)hipMemcpy(&h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    *cycles_per_mfma = (double)h / ((double)iters * (depth + 1) * wm * wn);
    return S2K_OK;
}
#endif
