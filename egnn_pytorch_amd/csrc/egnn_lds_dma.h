// LDS-DMA helpers shared by the edge-pass kernels (edge_fused.hip, edge_pw.hip): wave-level loads that land in LDS without a VGPR
// round trip.  Issued from inline asm on purpose -- with the builtins the compiler, which cannot tell LDS-DMA writes from the other
// LDS traffic, waits vmcnt(0) before every LDS access that follows; hidden from its counters the extra loads can only make its own
// vmcnt waits stricter (the counter retires in order), never weaker, and the kernels wait for their completion explicitly.
#pragma once
#include "egnn_common.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// One LDS-DMA instruction: lane l's 16 bytes at `g` (per-lane address) land at lds_base + 16 l (lds_base wave-uniform).
// Issued from inline asm on purpose, see the staging ring in edge_body.
__device__ __forceinline__ void lds_dma16(const char* g, char* lds_base)
{
#if defined(EGNN_EDGE_DMA_BUILTIN) && EGNN_EDGE_DMA_BUILTIN
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#else
    const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(m0) : "memory", "m0");
#endif
}

// The same with a wave-uniform base (scalar register pair) and a 32-bit per-lane byte offset: no 64-bit vector address arithmetic
// per instruction (the staging loops of the edge kernels issue a handful per chunk of the hidden dimension).
__device__ __forceinline__ void lds_dma16_s(const char* sbase, uint32_t voff, char* lds_base)
{
    const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(m0) : "memory", "m0");
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff)
{
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ uint32_t buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0);
}

// One gather instruction of the LDS-DMA path: lane l's 16 bytes at (descriptor base + voff + soff) land at lds_addr + 16 l
// (lds_addr, soff wave-uniform).  Inline asm for the same reason as lds_dma16.
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gather_dma16(u32x4s rsrc, uint32_t voff, uint32_t soff, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ u32x4s make_rsrc_words(const void* base, uint32_t bytes)
{
    const uint64_t a = (uint64_t)(size_t)base;
    u32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

