// lush-march: what the 64-points-per-wave chain kernels share (lush_mlp_wide.hip forward, lush_mlp_wide_bwd.hip backward):
// tile and ring geometry, the paired LDS-DMA, the per-wave stream context and the registers carried between passes.
#pragma once
#include "lush_common.h"
#include "lush_mlp.h"
#include "lush_mlp_dev.h"

#include <utility>

namespace lush {

constexpr int WD_MT = 256, WD_NT = 256;     // points per tile, threads (4 waves x 64 points)
#ifndef LUSH_WD_S
#define LUSH_WD_S 6
#endif
constexpr int WD_S = LUSH_WD_S;             // ring slots = prefetch distance in positions
constexpr int WD_SLOT = 8192;               // one position: 8 one-KiB fragments
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;


// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, E-1>)
template <int B, int E, class F>
__device__ __forceinline__ void wd_unroll(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        wd_unroll<B + 1, E>(f);
    }
}

// Two 1-KiB LDS-DMA pieces 4 KiB apart under one M0 save/restore with ONE per-lane offset register: two scalar bases.
// (The instruction's immediate offset is no help: measured in rounds 2 and 3, it does not move the global address of an
// LDS-DMA -- `offset:-4096` from base + 4096 fetched the wrong bytes.)
__device__ __forceinline__ void wd_dma_pair(const void* sbase0, const void* sbase1, unsigned voff, unsigned lds0, unsigned lds1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase0), "s"(sbase1), "s"(lds0), "s"(lds1) : "memory");
}

// a wave-uniform pointer the compiler may have placed in vector registers, back in scalar ones (inline-asm "s" operands)
__device__ __forceinline__ const char* wd_uniform(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

template <int N_>
__device__ __forceinline__ void wd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }


struct WdCtx {
#ifdef LUSH_PROF
    unsigned long long prof[16];
#endif
    const char* ring;      // LDS ring (generic pointer, fragment reads)
    unsigned ring_lds;     // its LDS byte address (DMA destination)
    const char* gbase;     // stream base (wave-uniform)
    unsigned slot_off;     // ring byte offset of the slot being consumed (0, SLOT, .. (S-1) SLOT)
    unsigned fetch_off;    // stream byte offset of the position the next refill DMA fetches (consumed position + S, wrapped)
    unsigned dma_base;     // ring_lds + 1 KiB x wave: LDS address of this wave's first DMA piece in slot 0
    int w, lane;
    unsigned voff;         // per-lane byte offset of this wave's first DMA piece of a position (the second: + 4 KiB)
};

struct WdCarry {             // registers that live from pass to pass
    bf16x8 a0[4];            // first-half fragments of the coming position
    u32x4 sb[2];             // stash rows read back from the LDS tile, waiting to be stored (two in flight)
};

}  // namespace lush
