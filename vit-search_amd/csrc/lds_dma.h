// LDS-DMA through a raw buffer descriptor (gfx950): one `buffer_load_dwordx4 ... offen lds` moves 64 lanes x 16 bytes from
// descriptor base + per-lane byte offset (VGPR, constant over a K loop) + scalar byte offset (advanced per slice) to 1 KB of LDS at
// M0.  Inline asm on purpose: hipcc waits vmcnt(0) in front of the first LDS read that follows the `raw_ptr_buffer_load_lds` BUILTIN
// (it cannot tell which LDS bytes the DMA writes), which drains every slice in flight; hidden from its bookkeeping the loads are
// counted by hand -- s_waitcnt vmcnt(N), then a barrier, then the reads (gemm_ntk.hip, gemm_nt_ln.hip).
#pragma once
#include "common.h"

namespace vr_dma {
typedef __attribute__((address_space(3))) char lds_char;
typedef int v4i __attribute__((ext_vector_type(4)));

// raw buffer descriptor: base, stride 0, num_records bytes (offsets >= num_records read as zero), 32-bit data format
__device__ __forceinline__ v4i make_rsrc(const void* ptr, unsigned num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)num_records);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(unsigned lds, unsigned voff, v4i rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
template <typename T> __device__ __forceinline__ unsigned lds_addr(T* p) { return (unsigned)(uintptr_t)(lds_char*)p; }
}  // namespace vr_dma
