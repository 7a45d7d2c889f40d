// Helpers shared by the LDS-DMA kernels (conv_dma.hip, conv_q4.hip): raw workgroup barrier, counted vmcnt wait, one 1-KiB DMA piece.
#pragma once
#include "common.h"

namespace s2k {

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One 1-KiB LDS-DMA piece: every lane moves 16 bytes from its own source offset to lds_dst + 16 * lane (lds_dst wave-uniform).
// The builtin is named in the DEVICE pass only: with it in a kernel body the host pass of hipcc (ROCm 7.2) drops that kernel's
// launch stub without a diagnostic (undefined symbol at load time).
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma16(rsrc_t r, float* lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)r; (void)lds_dst; (void)voff; (void)soff;
#endif
}


#if defined(S2K_TUNING) && defined(S2K_DMA_STAMPS)
// stamp builds (-DS2K_TUNING -DS2K_DMA_STAMPS; the flush is thousands of same-address atomics: ~50 us per launch, so kernel
// durations of such a build mean nothing): in-kernel stamps (s_memtime), summed over waves: {wave lifetime, set-up, wait + barrier, DMA issue, LDS reads + MFMAs,
// epilogue, waves, stages}
// (each kernel file owns its counters: device symbols do not link across translation units without -fgpu-rdc; the file defines
// DMA_DBG_SYM before including this header)
__device__ unsigned long long DMA_DBG_SYM[8];
#define DMA_STAMP() __builtin_amdgcn_s_memtime()
#define DMA_DBG_ADD(i, v) do { dbg_acc[i] += (unsigned long long)(v); } while (0)
#define DMA_DBG_DECL() unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DMA_DBG_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) if (dbg_acc[i_]) atomicAdd(&DMA_DBG_SYM[i_], dbg_acc[i_]); } while (0)
#else
#define DMA_STAMP() 0ull
#define DMA_DBG_ADD(i, v) do { } while (0)
#define DMA_DBG_DECL() do { } while (0)
#define DMA_DBG_FLUSH() do { } while (0)
#endif


}  // namespace s2k
