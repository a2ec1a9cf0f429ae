// lush-march: device helpers shared by the fused MLP kernels (lush_mlp.hip, lush_mlp_chain.hip).
#pragma once
#include "lush_common.h"
#include "lush_mlp.h"

namespace lush {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0), which
// would stall every layer on the acknowledgement of the in-flight stash stores.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ReLU decisions: per (32-point column block, mask layer, row-block) one 128-byte record = a 16-bit
// word per lane.  Canonical layout = the chain kernels' register layout: bit q of lane (col, h) is
// feature 16*(q>>3) + 8h + (q&7) of the 32-feature block ("activation > 0").  The column block is
// global (tile*CB + cb) so kernels with different tile sizes agree.  (Index is in 8-byte units.)
// Kernels whose accumulators hold the natural MFMA rows (bit 4g+e of lane (col, h) = feature
// 8g+4h+e) convert with mask_relayout(): half of each word comes from the partner lane (col, 1-h);
// the same formula maps either layout to the other.
__device__ __forceinline__ unsigned mask_relayout(unsigned own, unsigned other, int h) {
    const unsigned w0 = h ? other : own, w1 = h ? own : other;     // words of lanes (col, 0) and (col, 1)
    const int sh = 4 * h;
    return ((w0 >> sh) & 0xFu) | (((w1 >> sh) & 0xFu) << 4) | (((w0 >> (8 + sh)) & 0xFu) << 8) | (((w1 >> (8 + sh)) & 0xFu) << 12);
}
__device__ __forceinline__ long long mask_index(int tile, int n_ml, int ml, int nrb, int rb, int CB, int cb) {
    return ((((long long)tile * CB + cb) * n_ml + ml) * nrb + rb) * 16;
}

// ----------------------------------------------------------------------------
// positional encoding of one tile into the PE image
// ----------------------------------------------------------------------------
template <int NS, int DT>
__device__ __forceinline__ void pe_put(char* img, int plane_bytes, int row_bytes, int pt, int col, float v) {
    __bf16 p[NS];
    split_planes<NS, DT>(v, p);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        *reinterpret_cast<__bf16*>(img + s * plane_bytes + swz(pt, col >> 3, row_bytes) + (col & 7) * 2) = p[s];
}

// Point position exactly as the reference forms it: o + d*z, two roundings,
// no FMA (models/lushnerf.py:414, 525).
__device__ __forceinline__ void point_of(const float* __restrict__ rays, const float* __restrict__ z,
                                         int S, long long gpt, float (&x)[3], float (&d)[3]) {
    const long long ray = gpt / S;
    const float zz = z[gpt];
    const float* rr = rays + ray * 11;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] = __fadd_rn(rr[i], __fmul_rn(rr[3 + i], zz));
        d[i] = rr[8 + i];
    }
}

// sin and cos of the encoding arguments 2^k * coordinate (|x| up to a few thousand).  ocml's sincosf
// costs ~165 instructions per call because it carries the huge-argument (Payne-Hanek) path; this is
// the textbook form for bounded arguments: Cody-Waite reduction by pi/2 in three fused steps (the
// first is exact for |x| < 2^13: x - j*PI_A is a multiple of 2^-23 below 1), then the fdlibm float
// kernels on [-pi/4, pi/4].  Measured against float64 over 2^k * [-2, 2], k = 0..11: <= 1.5 ulp,
// <= 7.2e-8 absolute (tests/util: the reference's torch.sin on CPU and on CUDA differ by the same order).
__device__ __forceinline__ void lush_sincos(float x, float* sn, float* cs) {
    const float j = __builtin_rintf(__fmul_rn(x, 0.6366197723675814f));
    float r = __fmaf_rn(-j, 1.5707963705062866f, x);
    r = __fmaf_rn(-j, -4.371139000186243e-08f, r);
    r = __fmaf_rn(-j, -1.7151245100059906e-15f, r);
    const float z = __fmul_rn(r, r);
    float ps = __fmaf_rn(z, 0.0000027183114939898219064f, -0.000198393348360966317347f);
    ps = __fmaf_rn(z, ps, 0.0083333293858894631756f);
    ps = __fmaf_rn(z, ps, -0.166666666416265235595f);
    const float s = __fmaf_rn(__fmul_rn(z, r), ps, r);
    float pc = __fmaf_rn(z, 0.0000243904487962774090654f, -0.00138867637746099294692f);
    pc = __fmaf_rn(z, pc, 0.0416666233237390631894f);
    pc = __fmaf_rn(z, pc, -0.499999997251031003120f);
    const float c = __fmaf_rn(z, pc, 1.0f);
    const int q = (int)j;
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;       // quadrant: (s,c) (c,-s) (-s,-c) (-c,s)
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}

// Hardware sin / cos (v_sin_f32 / v_cos_f32 take revolutions) behind an exact range reduction: the coordinate over 2 pi
// as a double-float hi + lo (rev_split: one FMA), times the power of two (exact), v_fract (exact), + the scaled lo.
// 4.2e-7 absolute against float64 over 2^k [-2, 2], k = 0..9 (tools/micro/sincos_hw.hip): 1/500 of the fp16 grid -- used by
// the one-fp16-plane kernels only (the two-plane bf16 kernels carry 2^-17 and keep lush_sincos above); ~6 instructions
// instead of ~30.
__device__ __forceinline__ void rev_split(float x, float* hi, float* lo) {
    constexpr float C_HI = 0.15915494309189535f;
    constexpr float C_LO = (float)(0.15915494309189533576888 - (double)C_HI);
    *hi = x * C_HI;
    *lo = __builtin_fmaf(x, C_HI, -*hi) + x * C_LO;
}
__device__ __forceinline__ void sincos_rev(float hi, float lo, int k, float* sn, float* cs) {
    const float s = (float)(1 << k);
    const float r = __builtin_amdgcn_fractf(hi * s) + lo * s;
    *sn = __builtin_amdgcn_sinf(r);
    *cs = __builtin_amdgcn_cosf(r);
}

// (not inlined on purpose: the inlined sincosf bodies otherwise leave dozens of loop-invariant
// values live across the MFMA loops of the whole tile)
template <int NS, int MT, int NTHREADS, int DT>
__device__ __noinline__ void pe_tile(char* peimg, int plane_bytes, int row_bytes, const float* rays, const float* z,
                                        int S, int P, long long tile_pt0, int tid, const int* live = nullptr) {
    constexpr int PARTS = NTHREADS / MT;
    const int pt = tid % MT, part = tid / MT;
    const long long gpt = tile_pt0 + pt;
    float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (gpt < P) point_of(rays, z, S, live ? (long long)live[gpt] : gpt, x, d);      // (a live-point launch: its i-th point is grid point live[i])
    // units: 0..L_X-1 = frequency k of x; L_X..L_X+L_D-1 = frequency k of d; raw copies go with unit 0 / L_X
    for (int u = part; u < L_X + L_D; u += PARTS) {
        const bool isd = u >= L_X;
        const int k = isd ? u - L_X : u;
        const int base = isd ? PE_X : 0;
        const float f = (float)(1 << k);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = isd ? d[i] : x[i];
            if (k == 0) pe_put<NS, DT>(peimg, plane_bytes, row_bytes, pt, base + i, v);
            float s, c;
            lush_sincos(v * f, &s, &c);
            pe_put<NS, DT>(peimg, plane_bytes, row_bytes, pt, base + 3 + 6 * k + i, s);
            pe_put<NS, DT>(peimg, plane_bytes, row_bytes, pt, base + 3 + 6 * k + 3 + i, c);
        }
    }
    if (part == PARTS - 1) {   // zero padding columns that the K loops do read
        pe_put<NS, DT>(peimg, plane_bytes, row_bytes, pt, PE_X_VALID, 0.f);
#pragma unroll
        for (int c = PE_X + PE_D_VALID; c < PE_X + PE_D; ++c) pe_put<NS, DT>(peimg, plane_bytes, row_bytes, pt, c, 0.f);
    }
}

// One 1-KiB LDS-DMA: lane i moves 16 bytes from its own global address to LDS lds_base + 16*i.
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_base /* wave-uniform byte offset */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_base) : "memory");
}

// Same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: the address
// arithmetic of a stream of DMAs stays on the scalar unit (per-lane 64-bit addresses of every
// unrolled DMA are otherwise hoisted out of the tile loop and spilled).
__device__ __forceinline__ void dma16s(const void* sbase /* uniform */, unsigned voff, unsigned lds_base /* uniform */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

// The same for bytes that ONE workgroup reads ONCE (the weight-gradient kernel's stash stream): non-temporal policy
// (MI355X_MICROARCH.md, row nt-weights: issued -> landed -18 %).  Same-box A/B, fine pass, alternating processes (round 5,
// tools/ab_r05.sh): 4.093 / 4.098 ms against 4.189 / 4.196 ms with the default policy (-2.3 %).  The weight streams of the
// forward / chain kernels keep the default policy: every CU re-reads them from L2.
__device__ __forceinline__ void dma16s_stream(const void* sbase /* uniform */, unsigned voff, unsigned lds_base /* uniform */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

// Two of them under one M0 save/restore.
__device__ __forceinline__ void dma16s_x2(const void* sbase, unsigned voff0, unsigned lds0, unsigned voff1, unsigned lds1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff0), "s"(sbase), "s"(lds0), "v"(voff1), "s"(lds1) : "memory");
}

}  // namespace lush
