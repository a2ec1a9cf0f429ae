// lush-march: fused NeRF MLP kernels for gfx950 (MI355X).
//
//   mlp_fwd_kernel : positional encoding -> NL x (Linear+ReLU) with skip ->
//                    feature / alpha / views / rgb heads, one tile of MT points
//                    per workgroup pass, activations resident in LDS, weights
//                    streamed from L2 as pre-packed MFMA fragments.
//   mlp_bwd_kernel : the transposed chain dX = W^T dZ with ReLU masks, down to
//                    d/d(point) and d/d(viewdir) through the encoding.
//   dw_gemm_kernel : dW[o][i] = sum_p dZ[p][o] X[p][i] (split over points).
//
// Replaces Embedder.forward + NeRF.forward / NeRF_Noise.forward
// (utils/run_lushnerf_helpers.py:334-344, 394-423, 483-512) and
// NeRFAll.mlpforward (models/lushnerf.py:234-266).
//
// Orientation: features on MFMA rows (A = weights), points on MFMA columns
// (B = activations).  The accumulator of v_mfma_f32_32x32x16_bf16 then holds,
// per lane, ONE point (column) and 4 consecutive features in registers
// 4g..4g+3, so the next layer's B operand image [point][feature] is written
// with 8-byte ds_write_b64 and read back with 16-byte ds_read_b128.
//
// Precision: NS bf16 planes per operand (lush_common.h): NS=1 plain bf16,
// NS=2 ~2^-17 relative (parity mode), NS=3 ~fp32.  Accumulation is fp32.
#include "lush_common.h"
#include "lush_mlp.h"
#include "lush_mlp_dev.h"

namespace lush {

// fused MLP kernels take the wave count NW as a template parameter (8 = 2 waves per SIMD with a
// 256-register budget, 4 = 1 wave per SIMD with 512 registers for the 128-point tile)
constexpr int DW_THREADS = 256;           // weight-gradient GEMM / reductions: 4 waves as 2x2

// ----------------------------------------------------------------------------
// weight packing
// ----------------------------------------------------------------------------
// entry e of segment J -> its 64 x 8 fragment (NS planes) in dst
template <int NS, int DT>
__device__ __forceinline__ void pack_entry(const PackJob& J, int e, __bf16* __restrict__ dst) {
    const int kk = e / J.nrb, rb = e % J.nrb;
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int row = rb * 32 + (J.perm ? chain_row(r) : r);
    __bf16 out[NS][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kk * 16 + 8 * h + i;
        float v = 0.f;
        if (row < J.rows && k < J.cols) v = J.src[(long long)row * J.sr + (long long)k * J.sk];
        __bf16 p[NS];
        split_planes<NS, DT>(v, p);
#pragma unroll
        for (int s = 0; s < NS; ++s) out[s][i] = p[s];
    }
    __bf16* base = dst + ((long long)(J.dst_entry + e) * NS) * 64 * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = out[s][i];
        *reinterpret_cast<bf16x8*>(base + ((long long)s * 64 + lane) * 8) = v;
    }
}

template <int NS, int DT>
__global__ __launch_bounds__(64) void pack_kernel(const PackTable T, __bf16* __restrict__ dst) {
    int b = blockIdx.x;
    int j = 0;
    while (j + 1 < T.n && T.j[j + 1].first_block <= b) ++j;
    const PackJob J = T.j[j];
    pack_entry<NS, DT>(J, b - J.first_block, dst);
}

// fp32 block behind the fragments: biases + the two K<=3 head matrices (offsets: NetT::f32_*)
template <class N>
__device__ __forceinline__ float pack_f32_value(const MlpParams& P, int i) {
    float v = 0.f;
    if (i < N::f32_b_feat) v = P.b[i / N::HW][i % N::HW];
    else if (i < N::f32_b_alpha) v = P.b_feat[i - N::f32_b_feat];
    else if (i < N::f32_b_views) v = (i == N::f32_b_alpha) ? P.b_alpha[0] : 0.f;
    else if (i < N::f32_b_rgb) v = P.b_views[i - N::f32_b_views];
    else if (i < N::f32_w_rgb) v = (i - N::f32_b_rgb < 3) ? P.b_rgb[i - N::f32_b_rgb] : 0.f;
    else if (i < N::f32_w_alpha) v = P.w_rgb[i - N::f32_w_rgb];
    else v = P.w_alpha[i - N::f32_w_alpha];
    return v;
}
template <class N>
__global__ void pack_f32_kernel(const MlpParams P, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N::f32_total) return;
    dst[i] = pack_f32_value<N>(P, i);
}

// The whole plan in one launch: block b -> its job by binary search over the prefix sums.
__global__ __launch_bounds__(64) void pack_plan_kernel(const PlanHeader* __restrict__ H, float* __restrict__ zero_buf, long long zero_n) {
    // (the step's first launch also clears the flat gradient buffer: the trainer's zero_grad without a fill launch of its own)
    if (zero_buf != nullptr) {
        const long long t = (long long)blockIdx.x * 64 + threadIdx.x, nt = (long long)gridDim.x * 64;
        if ((reinterpret_cast<size_t>(zero_buf) & 15) == 0) {
            float4* z4 = reinterpret_cast<float4*>(zero_buf);
            for (long long i = t; i < zero_n / 4; i += nt) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long long i = zero_n / 4 * 4 + t; i < zero_n; i += nt) zero_buf[i] = 0.f;
        } else {
            for (long long i = t; i < zero_n; i += nt) zero_buf[i] = 0.f;
        }
    }
    const MlpParams* prm = reinterpret_cast<const MlpParams*>(H + 1);
    const PlanJob* jobs = reinterpret_cast<const PlanJob*>(prm + PLAN_MAX_NETS);
    const int b = blockIdx.x;
    int lo = 0, hi = H->n_jobs - 1;
    while (lo < hi) {                      // last job with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].j.first_block <= b) lo = mid; else hi = mid - 1;
    }
    const PlanJob& PJ = jobs[lo];
    const int e = b - PJ.j.first_block;
    if (PJ.kind == 0) {
        const PackJob J = PJ.j;
        __bf16* dst = reinterpret_cast<__bf16*>(PJ.dst);
        if (PJ.code == PLANES_F16) pack_entry<1, DT_F16>(J, e, dst);
        else if (PJ.code == 1) pack_entry<1, DT_BF16>(J, e, dst);
        else if (PJ.code == 2) pack_entry<2, DT_BF16>(J, e, dst);
        else pack_entry<3, DT_BF16>(J, e, dst);
    } else {
        const int i = e * 64 + threadIdx.x;
        float* dst = reinterpret_cast<float*>(PJ.dst);
        if (PJ.kind == 1) { if (i < NetNerf::f32_total) dst[i] = pack_f32_value<NetNerf>(prm[PJ.prm_index], i); }
        else if (i < NetNoise::f32_total) dst[i] = pack_f32_value<NetNoise>(prm[PJ.prm_index], i);
    }
}

// ----------------------------------------------------------------------------
// segment GEMM: acc[RB][CB] += Wseg[rows of this wave][K] * img[K][pts]
// ----------------------------------------------------------------------------
// Weight fragments go global(L2) -> registers through a ring of PF+1 slots (prefetch distance PF
// k-steps); activation fragments go LDS -> registers one k-step ahead through 2 slots.  The K loop
// is rolled in groups of G = PF+1 (even) steps so both ring indices stay static (runtime-indexed
// register arrays would go to scratch).  __builtin_amdgcn_sched_barrier pins "issue the prefetches,
// THEN the MFMAs of this step": without it hipcc (ROCm 7.2), under the 256-VGPR cap of a 512-thread
// workgroup, sinks every prefetch to just before its use and the loop runs load -> wait -> MFMA
// (measured: MFMA pipe 33 % busy, 56 % of wave cycles in s_waitcnt).
template <int NS, int RB, int CB, int KK, int DT = DT_BF16>
__device__ __forceinline__ void seg_gemm(f32x16 (&acc)[RB][CB], const bf16x8* __restrict__ wseg,
                                         int nrb, int rb0, const char* img, int plane_bytes,
                                         int row_bytes, int chunk0, int lane) {
    constexpr int PF = 5;
    constexpr int G = PF + 1;
    static_assert(G % 2 == 0, "B double buffer needs an even group");
    constexpr int NG = (KK >= 2 * PF + 1) ? (KK - 2 * PF - 1) / G + 1 : 0;
    const int r = lane & 31, h = lane >> 5;
    bf16x8 a[G][RB][NS];
    bf16x8 b[2][CB][NS];
    const bf16x8* wl = wseg + (long long)rb0 * NS * 64 + lane;
    const int kstride = nrb * NS * 64;
    auto loadA = [&](bf16x8 (&dst)[RB][NS], int kk) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int p = 0; p < NS; ++p) dst[i][p] = wl[(long long)kk * kstride + (i * NS + p) * 64];
    };
    auto loadB = [&](bf16x8 (&dst)[CB][NS], int kk) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int p = 0; p < NS; ++p)
                dst[cb][p] = *reinterpret_cast<const bf16x8*>(
                    img + p * plane_bytes + swz(cb * 32 + r, chunk0 + 2 * kk + h, row_bytes));
    };
    auto mma = [&](const bf16x8 (&as)[RB][NS], const bf16x8 (&bs)[CB][NS]) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
#ifdef LUSH_ABL_NOMFMA   // timing ablation only (wrong results): keep the operand loads, drop the MFMAs
#pragma unroll
                for (int p = 0; p < NS; ++p) asm volatile("" ::"v"(as[i][p]), "v"(bs[cb][p]));
#else
                acc[i][cb] = mfma_planes<NS, DT>(as[i], bs[cb], acc[i][cb]);
#endif
            }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < KK) loadA(a[s], s);
    loadB(b[0], 0);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int kb = g * G;
#pragma unroll
        for (int s = 0; s < G; ++s) {
            loadA(a[(s + PF) % G], kb + s + PF);
            loadB(b[(s + 1) & 1], kb + s + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a[s], b[s & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = NG * G; k < KK; ++k) {
        if (k + PF < KK) loadA(a[(k + PF) % G], k + PF);
        if (k + 1 < KK) loadB(b[(k + 1) & 1], k + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[k % G], b[k & 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int RB, int CB>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[RB][CB], const float* __restrict__ bias, int rb0,
                                         int nvalid, int h) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (rb0 + i) * 32 + acc_row(q, h);
            const float v = (bias != nullptr && row < nvalid) ? bias[row] : 0.f;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[i][cb][q] = v;
        }
}

// Write one 32x32 accumulator block as NS bf16 planes into the LDS image
// [pt][feature] (+ optionally the global stash and the ReLU sign bits).
template <int NS, bool RELU, int DT = DT_BF16>
__device__ __forceinline__ void store_block(const f32x16& acc, char* img, int plane_bytes, int row_bytes,
                                            int pt, int rb, int lane, __bf16* stash, long long stash_plane,
                                            int stash_ld, long long gpt, unsigned long long* mask_words) {
    const int h = lane >> 5;
#ifdef LUSH_ABL_NOEPI   // timing ablation only (wrong results)
    asm volatile("" ::"v"(acc));
    return;
#endif
    if (RELU && mask_words != nullptr) {
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) bits |= (acc[q] > 0.f ? 1u : 0u) << q;
        bits = mask_relayout(bits, (unsigned)__shfl_xor((int)bits, 32, 64), h);
        reinterpret_cast<unsigned short*>(mask_words)[lane] = (unsigned short)bits;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __bf16 pl[4][NS];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[4 * g + e];
            if (RELU) asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(v));   // plain max: fmaxf() adds a canonicalising v_max
            split_planes<NS, DT>(v, pl[e]);
        }
        const int f = rb * 32 + 8 * g + 4 * h;          // first of 4 consecutive features
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pl[e][p];
            *reinterpret_cast<bf16x4*>(img + p * plane_bytes + swz(pt, f >> 3, row_bytes) + (f & 7) * 2) = v;
        }
    }
}

// Copy `ncols` columns of every row of an LDS image to the global stash [plane][pt][ld] with
// 16-byte accesses: consecutive threads take consecutive chunks of a row, so each row leaves
// as full 128-byte lines (the image is de-swizzled on the way out).
template <int NS, int MT>
__device__ __noinline__ void copy_out(const char* img, int plane_bytes, int row_bytes, int ncols, __bf16* stash,
                                         long long stash_plane, int stash_ld, long long pt0, int tid, int nthreads,
                                         int planes = NS) {
    const int cpr = ncols >> 3;                    // 16-byte chunks per row
    for (int i = tid; i < planes * MT * cpr; i += nthreads) {
        const int c = i % cpr, pt = (i / cpr) % MT, p = i / (cpr * MT);
        const uint4 v = *reinterpret_cast<const uint4*>(img + p * plane_bytes + swz(pt, c, row_bytes));
#ifndef LUSH_NO_NT_STASH   // write-once stream: non-temporal stores (-2 % forward time)
        {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            u32x4 vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(stash + p * stash_plane + (pt0 + pt) * stash_ld + c * 8));
        }
#else
        *reinterpret_cast<uint4*>(stash + p * stash_plane + (pt0 + pt) * stash_ld + c * 8) = v;
#endif
    }
}

// Inlined form with static shapes: every LDS read of a plane is issued before its stores, nothing
// waits for the stores (a non-inlined callee drains vmcnt at entry and return, i.e. pays the full
// write-acknowledge latency once per layer).
template <int MT, int NCOLS, int NTHREADS>
__device__ __forceinline__ void copy_out_fast(const char* img, int plane_bytes, int row_bytes, __bf16* stash,
                                              long long stash_plane, long long pt0, int tid_in, int planes) {
    constexpr int CPR = NCOLS / 8;
    constexpr int PER = MT * CPR / NTHREADS;
    static_assert(MT * CPR % NTHREADS == 0, "copy_out_fast: ragged tile");
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    int tid = tid_in;
    asm volatile("" : "+v"(tid));
    for (int p = 0; p < planes; ++p) {
        u32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * NTHREADS, c = i % CPR, pt = i / CPR;
            v[j] = *reinterpret_cast<const u32x4*>(img + p * plane_bytes + swz(pt, c, row_bytes));
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * NTHREADS, c = i % CPR, pt = i / CPR;
            __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(stash + p * stash_plane + (pt0 + pt) * NCOLS + c * 8));
        }
    }
}

// Same, applying stored ReLU sign bits (backward) instead of computing them.
template <int NS>
__device__ __forceinline__ void store_block_masked(f32x16 acc, char* img, int plane_bytes, int row_bytes,
                                                   int pt, int rb, int lane, __bf16* stash,
                                                   long long stash_plane, int stash_ld, long long gpt,
                                                   const unsigned long long* mask_words) {
    if (mask_words != nullptr) {
        const unsigned short* mw = reinterpret_cast<const unsigned short*>(mask_words);
        const int bits = (int)mask_relayout(mw[lane], mw[lane ^ 32], lane >> 5);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((bits >> q) & 1)) acc[q] = 0.f;
        }
    }
    store_block<NS, false>(acc, img, plane_bytes, row_bytes, pt, rb, lane, stash, stash_plane, stash_ld, gpt,
                           nullptr);
}

// Re-derive lane / r / h from an opaque copy of the lane id at the top of every phase: everything
// computed from them (swizzled LDS addresses, fragment pointers, mask slots) then CANNOT be hoisted
// out of the layer loop by LICM and stays a short-lived temporary.  Without this ~150 VGPRs of
// hoisted address math were live across every GEMM loop of the tile and the 256-register budget of
// a 512-thread workgroup spilled inside the MFMA loops.
#define LUSH_FRESH_LANE()                       \
    int lane_f_ = lane;                         \
    asm volatile("" : "+v"(lane_f_));           \
    const int lane = lane_f_, r = lane & 31, h = lane >> 5; \
    (void)r; (void)h;

// ----------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------
template <class N, int NS, int MT, int NW, bool HAS_ALPHA, int DT>
__global__ __launch_bounds__(NW * 64) void mlp_fwd_kernel(const MlpFwdArgs A) {
    constexpr int NWAVES = NW, NTHREADS = NW * 64;
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL;
    constexpr int CB = MT / 32;
    constexpr int RB = N::NRB >= NWAVES ? N::NRB / NWAVES : 1;      // row-blocks per wave, trunk
    constexpr int RBV = N::NRBV >= NWAVES ? N::NRBV / NWAVES : 1;   // views layer
    constexpr int ACT_ROW = HW * 2, ACT_PLANE = MT * ACT_ROW;
    constexpr int PE_PLANE = MT * PE_ROW * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* actimg = smem;
    char* peimg = smem + NS * ACT_PLANE;
    float* alphabuf = reinterpret_cast<float*>(peimg + NS * PE_PLANE);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16x8* wpk = reinterpret_cast<const bf16x8*>(A.wpk);
    auto seg = [&](int entry) { return wpk + (long long)entry * NS * 64; };
    const float* f32 = reinterpret_cast<const float*>(A.wpk) + (long long)N::total_entries * NS * 256;
    const bool stash_on = A.write_stash != 0;

    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const long long pt0 = (long long)tile * MT;
        pe_tile<NS, MT, NTHREADS, DT>(peimg, PE_PLANE, PE_ROW * 2, A.rays, A.z, A.S, A.P, pt0, tid);
        lds_barrier();
        if (stash_on) {   // de-swizzled 16-byte copies of the 96 live PE columns
            for (int i = tid; i < A.stash_planes * MT * 12; i += NTHREADS) {
                const int c = i % 12, pt = (i / 12) % MT, p = i / (12 * MT);
                const uint4 v = *reinterpret_cast<const uint4*>(peimg + p * PE_PLANE + swz(pt, c, PE_ROW * 2));
                *reinterpret_cast<uint4*>(A.pe + p * A.plane_pe + (pt0 + pt) * PE_ROW + c * 8) = v;
            }
        }
        f32x16 acc[RB][CB];
        const int rb0 = w * RB;
        const bool trunk_active = (w * RB) < N::NRB;
        // ---- layer 0 ----
        if (trunk_active) {
            LUSH_FRESH_LANE();
            acc_bias<RB, CB>(acc, f32 + N::f32_b_trunk, rb0, HW, h);
            seg_gemm<NS, RB, CB, N::KKX, DT>(acc, seg(N::fwd_L(0, false)), N::NRB, rb0, peimg, PE_PLANE, PE_ROW * 2, 0, lane);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block<NS, true, DT>(acc[i][cb], actimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane,
                                          nullptr, 0, HW, pt0 + cb * 32 + r,
                                          stash_on ? A.mask + mask_index(tile, N::n_mask_layers, 0, N::NRB, rb0 + i, CB, cb) : nullptr);
        }
        lds_barrier();
        // Stash copies are issued AFTER the weight loads of the layer that consumes the image and
        // just before the barrier that lets it be overwritten: VMEM completes in order, so a store
        // in front of a weight load would add its acknowledgement latency to the load.
        // ---- layers 1 .. NL-1 ----
#pragma unroll 1
        for (int l = 1; l < NL; ++l) {
            LUSH_FRESH_LANE();
#ifdef LUSH_STASH_EARLY
            if (stash_on) copy_out_fast<MT, HW, NTHREADS>(actimg, ACT_PLANE, ACT_ROW, A.h0 + (l - 1) * A.h_stride, A.plane_h, pt0, tid, A.stash_planes);
#endif
            if (trunk_active) {
                acc_bias<RB, CB>(acc, f32 + N::f32_b_trunk + l * HW, rb0, HW, h);
                if (l == N::SKIP)
                    seg_gemm<NS, RB, CB, N::KKX, DT>(acc, seg(N::fwd_L(l, false)), N::NRB, rb0, peimg, PE_PLANE,
                                                 PE_ROW * 2, 0, lane);
                seg_gemm<NS, RB, CB, N::KKH, DT>(acc, seg(N::fwd_L(l, true)), N::NRB, rb0, actimg, ACT_PLANE, ACT_ROW, 0,
                                             lane);
            }
#ifndef LUSH_STASH_EARLY
            if (stash_on) copy_out_fast<MT, HW, NTHREADS>(actimg, ACT_PLANE, ACT_ROW, A.h0 + (l - 1) * A.h_stride, A.plane_h, pt0, tid, A.stash_planes);
#endif
            lds_barrier();
            if (trunk_active) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        store_block<NS, true, DT>(acc[i][cb], actimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane,
                                              nullptr, 0, HW, pt0 + cb * 32 + r,
                                              stash_on ? A.mask + mask_index(tile, N::n_mask_layers, l, N::NRB, rb0 + i, CB, cb) : nullptr);
            }
            lds_barrier();
        }
        // ---- feature (no activation) and alpha heads, both read h_{NL-1} ----
        {
        LUSH_FRESH_LANE();
        if (trunk_active) {
            acc_bias<RB, CB>(acc, f32 + N::f32_b_feat, rb0, HW, h);
            seg_gemm<NS, RB, CB, N::KKH, DT>(acc, seg(N::fwd_FEAT), N::NRB, rb0, actimg, ACT_PLANE, ACT_ROW, 0, lane);
        }
        if (HAS_ALPHA && w == NWAVES - 1) {
            f32x16 aa[1][CB];
            acc_bias<1, CB>(aa, f32 + N::f32_b_alpha, 0, 1, h);
            seg_gemm<NS, 1, CB, N::KKH, DT>(aa, seg(N::fwd_ALPHA), 1, 0, actimg, ACT_PLANE, ACT_ROW, 0, lane);
            if (h == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) alphabuf[cb * 32 + r] = aa[0][cb][0];
            }
        }
        if (stash_on) copy_out_fast<MT, HW, NTHREADS>(actimg, ACT_PLANE, ACT_ROW, A.h0 + (NL - 1) * A.h_stride, A.plane_h, pt0, tid, A.stash_planes);
        lds_barrier();
        if (trunk_active) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block<NS, false, DT>(acc[i][cb], actimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane,
                                           nullptr, 0, HW, pt0 + cb * 32 + r, nullptr);
        }
        lds_barrier();
        }
        // ---- views layer: relu(Wv [feature ; gamma(d)] + b) ----
        {
        LUSH_FRESH_LANE();
        f32x16 av[RBV][CB];
        const int rbv0 = w * RBV;
        const bool views_active = rbv0 < N::NRBV;
        if (views_active) {
            acc_bias<RBV, CB>(av, f32 + N::f32_b_views, rbv0, HV, h);
            seg_gemm<NS, RBV, CB, N::KKH, DT>(av, seg(N::fwd_VA), N::NRBV, rbv0, actimg, ACT_PLANE, ACT_ROW, 0, lane);
            seg_gemm<NS, RBV, CB, N::KKD, DT>(av, seg(N::fwd_VB), N::NRBV, rbv0, peimg, PE_PLANE, PE_ROW * 2, PE_X / 8,
                                          lane);
        }
        // (a 1- or 2-plane backward derives the feature layer's gradients from dZv^T h_{NL-1}: FeatFactorArgs)
        if (stash_on && A.stash_planes >= 3) copy_out_fast<MT, HW, NTHREADS>(actimg, ACT_PLANE, ACT_ROW, A.feat, A.plane_h, pt0, tid, A.stash_planes);
        lds_barrier();
        if (views_active) {
#pragma unroll
            for (int i = 0; i < RBV; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block<NS, true, DT>(av[i][cb], actimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rbv0 + i, lane,
                                          nullptr, 0, HV, pt0 + cb * 32 + r,
                                          stash_on ? A.mask + mask_index(tile, N::n_mask_layers, NL, N::NRB, rbv0 + i, CB, cb) : nullptr);
        }
        lds_barrier();
        }
        // ---- rgb head (3 rows) on wave 0; alpha joins from LDS ----
        if (w == 0) {
            LUSH_FRESH_LANE();
            f32x16 ar[1][CB];
            acc_bias<1, CB>(ar, f32 + N::f32_b_rgb, 0, 3, h);
            seg_gemm<NS, 1, CB, N::KKV, DT>(ar, seg(N::fwd_RGB), 1, 0, actimg, ACT_PLANE, ACT_ROW, 0, lane);
            if (h == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const long long gpt = pt0 + cb * 32 + r;
                    if (gpt < A.P) {
                        float4 o;
                        o.x = ar[0][cb][0];
                        o.y = ar[0][cb][1];
                        o.z = ar[0][cb][2];
                        o.w = HAS_ALPHA ? alphabuf[cb * 32 + r] : 0.f;
                        *reinterpret_cast<float4*>(A.raw + gpt * 4) = o;
                    }
                }
            }
        }
        if (stash_on) copy_out_fast<MT, HV, NTHREADS>(actimg, ACT_PLANE, ACT_ROW, A.hv, A.plane_hv, pt0, tid, A.stash_planes);
        lds_barrier();
    }
}

// ----------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------
constexpr int DPE_LD = 100;   // fp32 words per point in the d(gamma) scratch (96 used)

template <int CB>
__device__ __forceinline__ void dpe_add(float* dpe, const f32x16 (&acc)[1][CB], int rb, int col0, int lane,
                                        bool accumulate) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float* p = dpe + (cb * 32 + r) * DPE_LD + col0 + rb * 32 + 8 * g + 4 * h;
            f32x4 v = {acc[0][cb][4 * g], acc[0][cb][4 * g + 1], acc[0][cb][4 * g + 2], acc[0][cb][4 * g + 3]};
            if (accumulate) v += *reinterpret_cast<f32x4*>(p);
            *reinterpret_cast<f32x4*>(p) = v;
        }
}

template <class N, int NS, int MT, int NW, bool HAS_ALPHA>
__global__ __launch_bounds__(NW * 64) void mlp_bwd_kernel(const MlpBwdArgs A) {
    constexpr int NWAVES = NW, NTHREADS = NW * 64;
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL;
    constexpr int CB = MT / 32;
    constexpr int RB = N::NRB >= NWAVES ? N::NRB / NWAVES : 1;
    constexpr int RBV = N::NRBV >= NWAVES ? N::NRBV / NWAVES : 1;
    constexpr int ACT_ROW = HW * 2, ACT_PLANE = MT * ACT_ROW;
    constexpr int PARTS = NTHREADS / MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dimg = smem;
    float* dpe = reinterpret_cast<float*>(smem + NS * ACT_PLANE);
    float* drawbuf = dpe + MT * DPE_LD;                 // [MT][4]
    float* dxbuf = drawbuf + MT * 4;                    // [PARTS][MT][6]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16x8* wpk = reinterpret_cast<const bf16x8*>(A.wpk);
    auto seg = [&](int entry) { return wpk + (long long)entry * NS * 64; };
    const float* f32 = reinterpret_cast<const float*>(A.wpk) + (long long)N::total_entries * NS * 256;
    const float* w_rgb = f32 + N::f32_w_rgb;
    const float* w_alpha = f32 + N::f32_w_alpha;
    const int rb0 = w * RB, rbv0 = w * RBV;
    const bool trunk_active = rb0 < N::NRB, views_active = rbv0 < N::NRBV;

    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const long long pt0 = (long long)tile * MT;
        for (int i = tid; i < MT * 4; i += NTHREADS) {
            const long long gpt = pt0 + i / 4;
            drawbuf[i] = gpt < A.P ? A.draw[gpt * 4 + (i & 3)] : 0.f;
        }
        for (int i = tid; i < MT * DPE_LD; i += NTHREADS) dpe[i] = 0.f;
        lds_barrier();
        auto maskw = [&](int ml, int rb, int cb) {
            return A.mask + mask_index(tile, N::n_mask_layers, ml, N::NRB, rb, CB, cb);
        };
        // ---- dZv = (Wrgb^T d_rgb) * relu'(hv)   (K = 3: rank-3 update on the VALU) ----
        if (views_active) {
            LUSH_FRESH_LANE();
#pragma unroll
            for (int i = 0; i < RBV; ++i) {
                f32x16 a[CB];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (rbv0 + i) * 32 + acc_row(q, h);
                    const float w0 = w_rgb[row], w1 = w_rgb[HV + row], w2 = w_rgb[2 * HV + row];
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        const float* dr = drawbuf + (cb * 32 + r) * 4;
                        a[cb][q] = w0 * dr[0] + w1 * dr[1] + w2 * dr[2];
                    }
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block_masked<NS>(a[cb], dimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rbv0 + i, lane, A.dzv,
                                           A.plane_hv, HV, pt0 + cb * 32 + r, maskw(NL, rbv0 + i, cb));
            }
        }
        lds_barrier();
        // ---- d_feature = Wva^T dZv ; d gamma(d) = Wvb^T dZv ----
        f32x16 acc[RB][CB];
        {
        LUSH_FRESH_LANE();
        if (trunk_active) {
            acc_bias<RB, CB>(acc, nullptr, rb0, 0, h);
            seg_gemm<NS, RB, CB, N::KKV>(acc, seg(N::bwd_VAT), N::NRB, rb0, dimg, ACT_PLANE, ACT_ROW, 0, lane);
        }
        if (w == NWAVES - 1) {
            f32x16 ad[1][CB];
            acc_bias<1, CB>(ad, nullptr, 0, 0, h);
            seg_gemm<NS, 1, CB, N::KKV>(ad, seg(N::bwd_VBT), 1, 0, dimg, ACT_PLANE, ACT_ROW, 0, lane);
            dpe_add<CB>(dpe, ad, 0, PE_X, lane, false);
        }
        copy_out_fast<MT, HV, NTHREADS>(dimg, ACT_PLANE, ACT_ROW, A.dzv, A.plane_hv, pt0, tid, NS);
        lds_barrier();
        if (trunk_active) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block_masked<NS>(acc[i][cb], dimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane, A.dfeat,
                                           A.plane_h, HW, pt0 + cb * 32 + r, nullptr);
        }
        lds_barrier();
        }
        // ---- dZ_{NL-1} = (Wfeat^T d_feature + Walpha^T d_alpha) * relu'(h_{NL-1}) ----
        {
        LUSH_FRESH_LANE();
        if (trunk_active) {
            acc_bias<RB, CB>(acc, nullptr, rb0, 0, h);
            seg_gemm<NS, RB, CB, N::KKH>(acc, seg(N::bwd_FEATT), N::NRB, rb0, dimg, ACT_PLANE, ACT_ROW, 0, lane);
            if (HAS_ALPHA) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float wa = w_alpha[(rb0 + i) * 32 + acc_row(q, h)];
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) acc[i][cb][q] += wa * drawbuf[(cb * 32 + r) * 4 + 3];
                    }
            }
        }
        copy_out_fast<MT, HW, NTHREADS>(dimg, ACT_PLANE, ACT_ROW, A.dfeat, A.plane_h, pt0, tid, NS);
        lds_barrier();
        if (trunk_active) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    store_block_masked<NS>(acc[i][cb], dimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane,
                                           A.dz0 + (NL - 1) * A.dz_stride, A.plane_h, HW, pt0 + cb * 32 + r, maskw(NL - 1, rb0 + i, cb));
        }
        lds_barrier();
        }
        // ---- trunk: dZ_{l-1} = (W_l^T dZ_l) * relu'(h_{l-1}) ----
#pragma unroll 1
        for (int l = NL - 1; l >= 1; --l) {
            LUSH_FRESH_LANE();
            if (l == N::SKIP && w < 2) {    // gamma(x) rows of the skip layer's input
                f32x16 ap[1][CB];
                acc_bias<1, CB>(ap, nullptr, 0, 0, h);
                seg_gemm<NS, 1, CB, N::KKH>(ap, seg(N::bwd_LT(l, false)), 2, w, dimg, ACT_PLANE, ACT_ROW, 0, lane);
                dpe_add<CB>(dpe, ap, w, 0, lane, false);
            }
            if (trunk_active) {
                acc_bias<RB, CB>(acc, nullptr, rb0, 0, h);
                seg_gemm<NS, RB, CB, N::KKH>(acc, seg(N::bwd_LT(l, true)), N::NRB, rb0, dimg, ACT_PLANE, ACT_ROW, 0,
                                             lane);
            }
            copy_out_fast<MT, HW, NTHREADS>(dimg, ACT_PLANE, ACT_ROW, A.dz0 + l * A.dz_stride, A.plane_h, pt0, tid, NS);
            lds_barrier();
            if (trunk_active) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        store_block_masked<NS>(acc[i][cb], dimg, ACT_PLANE, ACT_ROW, cb * 32 + r, rb0 + i, lane,
                                               A.dz0 + (l - 1) * A.dz_stride, A.plane_h, HW, pt0 + cb * 32 + r, maskw(l - 1, rb0 + i, cb));
            }
            lds_barrier();
        }
        // ---- layer 0: d gamma(x) += W_0^T dZ_0 ----
        if (w < 2) {
            LUSH_FRESH_LANE();
            f32x16 ap[1][CB];
            acc_bias<1, CB>(ap, nullptr, 0, 0, h);
            seg_gemm<NS, 1, CB, N::KKH>(ap, seg(N::bwd_LT(0, false)), 2, w, dimg, ACT_PLANE, ACT_ROW, 0, lane);
            dpe_add<CB>(dpe, ap, w, 0, lane, N::SKIP >= 0);
        }
        copy_out_fast<MT, HW, NTHREADS>(dimg, ACT_PLANE, ACT_ROW, A.dz0, A.plane_h, pt0, tid, NS);
        lds_barrier();
        // ---- through the encoding: d/dx_i = g[i] + sum_k 2^k (cos(2^k x_i) g_sin - sin(2^k x_i) g_cos) ----
        {
            const int pt = tid % MT, part = tid / MT;
            const long long gpt = pt0 + pt;
            float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
            if (gpt < A.P) point_of(A.rays, A.z, A.S, gpt, x, d);
            float gx[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
            const float* g = dpe + pt * DPE_LD;
            for (int u = part; u < L_X + L_D; u += PARTS) {
                const bool isd = u >= L_X;
                const int k = isd ? u - L_X : u;
                const int base = isd ? PE_X : 0;
                const float f = (float)(1 << k);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float v = isd ? d[i] : x[i];
                    float s, c;
                    lush_sincos(v * f, &s, &c);
                    float t = f * (c * g[base + 3 + 6 * k + i] - s * g[base + 3 + 6 * k + 3 + i]);
                    if (k == 0) t += g[base + i];
                    if (isd) gd[i] += t; else gx[i] += t;
                }
            }
            float* o = dxbuf + (part * MT + pt) * 6;
            o[0] = gx[0]; o[1] = gx[1]; o[2] = gx[2]; o[3] = gd[0]; o[4] = gd[1]; o[5] = gd[2];
        }
        lds_barrier();
        for (int i = tid; i < MT * 6; i += NTHREADS) {
            const int pt = i / 6, c = i % 6;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < PARTS; ++p) s += dxbuf[(p * MT + pt) * 6 + c];
            const long long gpt = pt0 + pt;
            if (gpt < A.P) A.dpts[gpt * 8 + (c < 3 ? c : c + 1)] = s;
        }
        lds_barrier();
    }
}

// ----------------------------------------------------------------------------
// weight gradient: dW[o][i] += sum_p dZ[p][o] * X[p][i]
// ----------------------------------------------------------------------------
// One workgroup owns the FULL 256 x 256 output (so every stashed byte is read from HBM exactly
// once per layer) for a slice of the points; 8 waves as 2 (o) x 4 (i), wave tile 128 x 64.
// Both operands are [point][feature] row-major in HBM; the MFMA wants 8 consecutive POINTS per
// lane, so fragments come out of the LDS tile through the transposing ds_read_b64_tr_b16
// (cdna_hip_programming.md T10).  LDS rows are padded to 576 B: the 4 rows x 64 B that one
// half-wave touches then fall on 64 distinct banks.  Tiles of 32 points are double buffered:
// global -> registers for tile t+1 is issued before the MFMAs of tile t (T14), written to the
// other buffer afterwards, one barrier per tile.
constexpr int DW_T = 256, DW_ROW = DW_T * 2 + 64;
template <int NS> struct DwKt { static constexpr int v = NS == 3 ? 16 : 32; };   // 3 planes: 16-point tiles fit 160 KB
constexpr int DW_THREADS2 = 512;

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int col0, int lane) {
    // lane supplies the address of row (k0 + 8*(G>>1) + 4t + q), columns col0 + 16*(G&1) + 4p .. +3
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a0 = tile + (k0 + 8 * (G >> 1) + q) * DW_ROW + (col0 + 16 * (G & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * DW_ROW));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.v;
}

// fp16 X stash (written by the fp16 forward) -> bf16 fragment (the gradients are bf16: fp16's range is
// unsafe for dZ); 8 elements, two conversions each
__device__ __forceinline__ bf16x8 f16_frag_to_bf16(bf16x8 v) {
    const f16x8 hv = __builtin_bit_cast(f16x8, v);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(float)hv[e];
    return o;
}

template <int NS, bool XF16>
__global__ __launch_bounds__(DW_THREADS2) void dw_gemm_kernel(const DwArgs A) {
    constexpr int DW_KT = DwKt<NS>::v;
    extern __shared__ __attribute__((aligned(16))) char tiles[];   // [2 buffers][Z planes | X planes][KT][DW_ROW]
    constexpr int PLANE = DW_KT * DW_ROW, OPER = NS * PLANE, BUF = 2 * OPER;
    constexpr int LPT = NS * DW_KT * (DW_T / 8) / DW_THREADS2;     // 16-byte loads per thread and operand
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wo = w >> 2, wi = w & 3;
    const long long p_begin = (long long)blockIdx.x * A.pts_per_split;
    long long p_end = p_begin + A.pts_per_split;
    if (p_end > A.Ppad) p_end = A.Ppad;
    const int n_tiles = (int)((p_end - p_begin) / DW_KT);
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    const bool do_bias = A.db != nullptr;
    f32x16 accb;
#pragma unroll
    for (int q = 0; q < 16; ++q) accb[q] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    // waves whose whole column range is padding do no MFMA work
    const bool wave_live = (wo * 128 < A.n_out) && (wi * 64 < A.k_in);
    const bool bias_only = !wave_live && do_bias && (wo * 128 < A.n_out);
    // Register staging with TWO tiles in flight when the staging set is small (1 plane: 16 VGPRs per
    // tile): loads of tile t+2 are issued before the MFMAs of tile t, tile t+1 is written to the other
    // LDS buffer after them.  One tile in flight per CU pulled only ~3.9 TB/s from HBM.
    constexpr int SETS = NS == 1 ? 2 : 1;
    uint4 rz[SETS][LPT], rx[SETS][LPT];
    auto gload = [&](uint4 (&dz)[LPT], uint4 (&dx)[LPT], long long p0) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int i = tid + j * DW_THREADS2;
            const int c = i & 31, row = (i >> 5) % DW_KT, pl = i / (32 * DW_KT);
            dz[j] = make_uint4(0, 0, 0, 0);
            dx[j] = make_uint4(0, 0, 0, 0);
            if (c * 8 < A.n_out)
                dz[j] = *reinterpret_cast<const uint4*>(A.Z + pl * A.z_plane + (p0 + row) * A.ldz + c * 8);
            if (c * 8 < A.k_in)
                dx[j] = *reinterpret_cast<const uint4*>(A.X + pl * A.x_plane + (p0 + row) * A.ldx + A.xcol0 + c * 8);
        }
    };
    auto lstore = [&](const uint4 (&dz)[LPT], const uint4 (&dx)[LPT], int buf) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int i = tid + j * DW_THREADS2;
            const int c = i & 31, row = (i >> 5) % DW_KT, pl = i / (32 * DW_KT);
            char* base = tiles + buf * BUF + pl * PLANE + row * DW_ROW + c * 16;
            *reinterpret_cast<uint4*>(base) = dz[j];
            *reinterpret_cast<uint4*>(base + OPER) = dx[j];
        }
    };
    auto compute = [&](int buf) {
        if (wave_live) {
            const char* zt = tiles + buf * BUF;
            const char* xt = zt + OPER;
#pragma unroll
            for (int ks = 0; ks < DW_KT / 16; ++ks) {
                bf16x8 a[4][NS], b[2][NS];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u][pl] = tr_frag(zt + pl * PLANE, ks * 16, wo * 128 + u * 32, lane);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        b[u][pl] = tr_frag(xt + pl * PLANE, ks * 16, wi * 64 + u * 32, lane);
                        if constexpr (XF16) b[u][pl] = f16_frag_to_bf16(b[u][pl]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) acc[u][v] = mfma_planes<NS>(a[u], b[v], acc[u][v]);
                if (do_bias) {   // db[o] = sum_p dZ[p][o]: MFMA against an all-ones B; wave (wo, wi) sums block wi
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) {
                        const bf16x8 sel = wi == 0 ? a[0][pl] : (wi == 1 ? a[1][pl] : (wi == 2 ? a[2][pl] : a[3][pl]));
                        accb = mfma_bf16(sel, ones, accb);
                    }
                }
            }
        } else if (bias_only) {   // column range is padding: this wave still owns bias block wi
            const char* zt = tiles + buf * BUF;
#pragma unroll
            for (int ks = 0; ks < DW_KT / 16; ++ks)
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    accb = mfma_bf16(tr_frag(zt + pl * PLANE, ks * 16, wo * 128 + wi * 32, lane), ones, accb);
        }
    };
    auto tile_p0 = [&](int t) { return p_begin + (long long)t * DW_KT; };
    if constexpr (SETS == 1) {
        if (n_tiles > 0) {
            gload(rz[0], rx[0], tile_p0(0));
            lstore(rz[0], rx[0], 0);
        }
        lds_barrier();   // LDS-only: must not drain the tiles still in flight
        for (int t = 0; t < n_tiles; ++t) {
            if (t + 1 < n_tiles) gload(rz[0], rx[0], tile_p0(t + 1));
            compute(t & 1);
            if (t + 1 < n_tiles) lstore(rz[0], rx[0], (t & 1) ^ 1);
            lds_barrier();   // LDS-only: must not drain the tiles still in flight
        }
    } else {
        // tile t lives in LDS buffer t&1; register set s holds tile with (t & 1) == s
        if (n_tiles > 0) {
            gload(rz[0], rx[0], tile_p0(0));
            if (n_tiles > 1) gload(rz[1], rx[1], tile_p0(1));
            lstore(rz[0], rx[0], 0);
        }
        lds_barrier();   // LDS-only: must not drain the tiles still in flight
        for (int t = 0; t < n_tiles; t += 2) {
            // even step: set 0 is free (tile t is already in LDS), set 1 holds tile t+1
            if (t + 2 < n_tiles) gload(rz[0], rx[0], tile_p0(t + 2));
            compute(0);
            if (t + 1 < n_tiles) lstore(rz[1], rx[1], 1);
            lds_barrier();   // LDS-only: must not drain the tiles still in flight
            if (t + 1 >= n_tiles) break;
            // odd step: set 1 is free, set 0 holds tile t+2
            if (t + 3 < n_tiles) gload(rz[1], rx[1], tile_p0(t + 3));
            compute(1);
            if (t + 2 < n_tiles) lstore(rz[0], rx[0], 0);
            lds_barrier();   // LDS-only: must not drain the tiles still in flight
        }
    }
    const int r = lane & 31, h = lane >> 5;
    if (do_bias && r == 0 && (wave_live || bias_only)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int o = wo * 128 + wi * 32 + acc_row(q, h);
            if (o < A.n_out) atomicAdd(A.db + o, accb[q]);
        }
    }
    if (!wave_live) return;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = wo * 128 + u * 32 + acc_row(q, h);
                const int i = wi * 64 + v * 32 + r;
                if (o < A.n_out && i < A.k_in) atomicAdd(A.dW + (long long)o * A.ldw + A.wcol0 + i, acc[u][v][q]);
            }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA ring of the grouped kernel below.  Round 1's register-staged kernel (MFMAs removed: 6.41 vs 6.54 ms) was
// paced by the global -> VGPR -> LDS path at ~4.1 TB/s with 64 KB in flight per CU.  Here tiles go HBM -> LDS directly
// (global_load_lds_dwordx4, no VGPRs, no ds_write) into a 4-stage ring, so 3 tiles are in flight while one is consumed.
//  * DMA writes LDS linearly (M0 base + lane*16), so rows cannot be padded; the bank-conflict fix for
//    the transposing reads is an XOR swizzle of the 16-byte chunk index with (row & 3) << 2, applied
//    to the per-lane SOURCE address and again on the read (cdna_hip_programming.md rule 21).
//  * Columns beyond the valid width are never written (lanes masked): the ring is zeroed per job, and a
//    given (row, position) is either always or never written, so it stays zero.
//  * Completion: each wave waits its own DMAs with a counted s_waitcnt vmcnt(4 * younger stages),
//    then ONE raw s_barrier per tile makes every wave's rows visible; the slot refilled after that
//    barrier was last read in the previous iteration, which every wave has left.
#ifndef LUSH_DW_STAGES
#define LUSH_DW_STAGES 4
#endif
constexpr int DMA_STAGES = LUSH_DW_STAGES, DMA_KT = 32, DMA_ROWB = 512, DMA_OPER = DMA_KT * DMA_ROWB;

// bf16 X stash (the hi plane of a 2-plane forward) -> fp16 fragment for the fp16 gradient GEMM: exact (8-bit
// mantissa into 11 bits) unless |x| < 2^-24 (flushed; such an activation contributes nothing) or > 65504 (no
// activation of these networks gets near: the encoding is in [-1, 1], weights ~1/16)
__device__ __forceinline__ bf16x8 bf16_frag_to_f16(bf16x8 v) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(float)v[e];
    return __builtin_bit_cast(bf16x8, o);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight-gradient launch: ONE launch per network pass instead of one per layer.
// Measured on the per-layer kernel above (fine / coarse trunk layers, 2.68 / 1.34 GB each): 500 / 295 us, i.e. an
// asymptotic 6.5 TB/s plus ~90 us of fixed cost per launch -- every workgroup drains its 256 KB of fp32 atomics at
// the same moment (64 MB per launch at the chip's ~1.3 TB/s atomic rate) while nothing streams, then the next launch
// refills its ring from cold.  Here a persistent workgroup owns one slice of the points and walks the layer list;
// workgroup b starts at job b mod n, so at any moment the workgroups of the chip are spread over all layers: the
// atomics of one overlap the streaming of the others and no two neighbours add into the same matrix at once.
// A job may carry a second input block X2 that shares its dZ (skip layer: gamma(x) | h; views layer: feature |
// gamma(d)): the per-layer launches read that dZ twice (7 GB per step).
//   stage = [Z 32 x 512 B | X 32 x 512 B | X2 32 x 128 B]; X2 is swizzled chunk ^= ((row >> 1) & 1) << 2 (rows of 128 B:
//   the four rows of a transposing read then sit on four different 16-bank groups).
// Two bf16 planes per operand (NS = 2, the strict mode): the 32 rows of a tile are [plane 0: 16 points | plane 1: the
// same 16 points], i.e. one k-step per stage with the three plane products hi*hi + hi*lo + lo*hi; a wave's DMA rows lie
// in one plane, so the plane offset is part of its scalar base and nothing else changes.
constexpr int GRP_X2_ROWB = 128, GRP_X2 = DMA_KT * GRP_X2_ROWB, GRP_STAGE = 2 * DMA_OPER + GRP_X2;

// Byte offset (inside an operand tile) of the first of the two transposing reads of a fragment at column col0; the
// second sits 4 rows below, k-step ks 16 rows below: both plain immediates.
__device__ __forceinline__ unsigned tr_off_sw(int col0, int lane) {
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (G >> 1) + q;
    const int col = col0 + 16 * (G & 1) + 4 * p;
    return (unsigned)(row * DMA_ROWB + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2);
}
__device__ __forceinline__ unsigned tr_off_x2(int col0, int lane) {
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (G >> 1) + q;
    const int col = col0 + 16 * (G & 1) + 4 * p;
    return (unsigned)(row * GRP_X2_ROWB + ((((col >> 3) ^ (((row >> 1) & 1) << 2))) << 4) + (col & 7) * 2);
}
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_read(const char* a0) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ROWB));
    return u.v;
}

template <int N_>
__device__ __forceinline__ void grp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Everything a workgroup needs of a job while it streams: wave-uniform scalars and the per-lane constants of its DMAs.
struct GrpStream {
    const char* src;        // wave-uniform: this wave's operand (Z for waves 0..3, X for 4..7) at the slice's first point
    unsigned stride;        // bytes per 32-point tile of that operand
    unsigned voff[4];       // per-lane byte offset of the wave's four 1-KiB pieces (two rows each) inside a tile
    unsigned dst[4];        // wave-uniform LDS byte offset of those pieces inside a stage
    bool on[4];             // lane moves a chunk that exists (columns < the job's width)
    const char* src2;       // X2 (waves 0..3 when the job has one)
    unsigned stride2, voff2, dst2;
    bool on2;
    // X2 re-encoded instead of read (DwGroup::xd, DwJob::pe_mode): this lane's 16-byte chunk of gamma(.) of its tile row
    const char* xd;         // uniform; null: X2 is read.  Else: the slice's first point's quadruple (x or viewdir) in DwGroup::xd
    long long pe_last;      // last valid point of the array relative to the slice's first (reads are clamped to it)
    unsigned pe_row16;      // per-lane (waves 0..3): 16 x the lane's tile row = its byte offset inside a tile's 512 bytes of quadruples
    unsigned pe_dst;        // per-lane LDS byte offset of the chunk inside a stage
    int pe_gch, pe_nvalid;  // source chunk (8 columns) of gamma(.), its number of valid columns (63 / 27)
};
// Quadruples (16 bytes per point) of GRP_PE_CHUNK points at a time through two LDS buffers behind the stages: one LDS-DMA per wave
// and chunk, issued a whole chunk (16 tiles) ahead, so the ordinary ring waits cover it and no wait of its own stands in the loop.
constexpr int GRP_PE_CHUNK = 512, GRP_PE_BYTES = GRP_PE_CHUNK * 16;

// 8 columns [8 gch, 8 gch + 8) of gamma(c) = [c, sin(2^k c), cos(2^k c) ...] (utils/run_lushnerf_helpers.py:334-361) as fp16 to
// `dst` (LDS), bit for bit what wd_pe_tile (lush_mlp_wide.hip) wrote into the forward's LDS image: the same range reduction, the
// same hardware sin / cos, the same rounding.  Columns >= nvalid are zero.  One pair of columns per trip of a ROLLED loop: unrolled,
// the compiler evaluated the 40 lane predicates of the 8 columns up front (106 scalar registers spilled).
__device__ __forceinline__ void dw_pe_write(char* dst, const f32x4 c, int gch, int nvalid) {
    float hi[3], lo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rev_split(c[i], &hi[i], &lo[i]);
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        float v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 8 * gch + 2 * p + t;
            const int q = j < 3 ? 0 : j - 3;
            const int k = (q * 43) >> 8;                 // q / 6 for 0 <= q < 64
            const int r = q - 6 * k;
            const int i = j < 3 ? j : (r < 3 ? r : r - 3);
            const float ci = i == 0 ? c[0] : (i == 1 ? c[1] : c[2]);
            const float h_ = i == 0 ? hi[0] : (i == 1 ? hi[1] : hi[2]);
            const float l_ = i == 0 ? lo[0] : (i == 1 ? lo[1] : lo[2]);
            const float s = __builtin_ldexpf(1.0f, k);
            const float rr = __builtin_amdgcn_fractf(h_ * s) + l_ * s;
            const float sc = r < 3 ? __builtin_amdgcn_sinf(rr) : __builtin_amdgcn_cosf(rr);
            v[t] = j < 3 ? ci : (j < nvalid ? sc : 0.f);
        }
        unsigned w_[1];
        split_pair<1, DT_F16>(v[0], v[1], w_);
        *reinterpret_cast<unsigned*>(dst + 4 * p) = w_[0];
    }
}

// The same for 4 columns [8 gch + 4 half, .. + 4): 8 bytes per lane, so that ALL eight waves share a tile's 32 rows x 64 columns
// (16 lanes per row).  With four waves re-encoding 16 bytes per lane the other four idled behind them at the stage's barrier and
// a workgroup spent 27 ns per point in its three re-encoding jobs (profiles/r05_dw_jobs.md) -- the jobs ran 1.3 x .. 1.9 x their
// streaming time.
__device__ __forceinline__ void dw_pe_write4(char* dst, const f32x4 c, int gch, int half, int nvalid) {
    float hi[3], lo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rev_split(c[i], &hi[i], &lo[i]);
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        float v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 8 * gch + 4 * half + 2 * p + t;
            const int q = j < 3 ? 0 : j - 3;
            const int k = (q * 43) >> 8;                 // q / 6 for 0 <= q < 64
            const int r = q - 6 * k;
            const int i = j < 3 ? j : (r < 3 ? r : r - 3);
            const float ci = i == 0 ? c[0] : (i == 1 ? c[1] : c[2]);
            const float h_ = i == 0 ? hi[0] : (i == 1 ? hi[1] : hi[2]);
            const float l_ = i == 0 ? lo[0] : (i == 1 ? lo[1] : lo[2]);
            const float s = __builtin_ldexpf(1.0f, k);
            const float rr = __builtin_amdgcn_fractf(h_ * s) + l_ * s;
            const float sc = r < 3 ? __builtin_amdgcn_sinf(rr) : __builtin_amdgcn_cosf(rr);
            v[t] = j < 3 ? ci : (j < nvalid ? sc : 0.f);
        }
        unsigned w_[1];
        split_pair<1, DT_F16>(v[0], v[1], w_);
        *reinterpret_cast<unsigned*>(dst + 4 * p) = w_[0];
    }
}

// The streaming loop of one job, specialised on the number of 32-column X2 blocks (0: none, the side accumulator is the
// bias alone).  Nothing in it depends on the job except through `st` (registers) and three wave-uniform flags.
LUSH_CLOCK_DECL(lush_clock_dw)
#ifdef LUSH_PROF_DW   // developer build: s_memtime counts of workgroup 0 / thread 0 per phase of a job, read back through lush_debug_prof_dw
__device__ unsigned long long lush_prof_dw[16];
__device__ unsigned long long lush_prof_dw_span[2 * 1024];      // [b] start, [1024 + b] end of workgroup b (s_memtime: one clock for the chip)
#define DPROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DPROF_ADD(slot, t0) do { if (blockIdx.x == 0 && threadIdx.x == 0) lush_prof_dw[slot] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#else
#define DPROF_T(var)
#define DPROF_ADD(slot, t0)
#endif
template <bool XF16, bool ZF16, int NV2, int NS>
__device__ __forceinline__ void grp_stream(const GrpStream& st, const char* tiles, unsigned lds0, int n_tiles, int w, int lane,
                                           bool wave_live, bool row_live, bool x2_wave, const unsigned (&a_off)[4],
                                           const unsigned (&b_off)[2], unsigned sel_off, const unsigned (&x2_off)[2],
                                           f32x16 (&acc)[4][2], f32x16 (&accs)[2]) {
    auto mm = [](bf16x8 a, bf16x8 b, f32x16 c) { return ZF16 ? mfma_f16(a, b, c) : mfma_bf16(a, b, c); };
    auto xcv = [](bf16x8 v) {
        if constexpr (XF16 && !ZF16) return f16_frag_to_bf16(v);
        else if constexpr (!XF16 && ZF16) return bf16_frag_to_f16(v);
        else return v;
    };
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = ZF16 ? __builtin_bit_cast(__bf16, (_Float16)1.0f) : (__bf16)1.0f;
    const bool bias_lane = (lane & 31) == 31;
    const char* src = st.src;
    const char* src2 = st.src2;
    constexpr int KT_ = DMA_KT / NS;
    const char* coords = tiles + DMA_STAGES * GRP_STAGE;         // [2][GRP_PE_BYTES] (allocated when DwGroup::xd is given)
    int ti = 0;                                                   // tiles issued
    auto pe_chunk = [&](int k) {      // this wave's 64 quadruples of chunk k -> LDS buffer k & 1 (one LDS-DMA, gathered: 32-byte stride)
        long long pt = (long long)k * GRP_PE_CHUNK + w * 64 + lane;
        if (pt > st.pe_last) pt = st.pe_last;
        dma16s(st.xd, (unsigned)(pt * 32), __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(DMA_STAGES * GRP_STAGE + (k & 1) * GRP_PE_BYTES + w * 1024)));
    };
    constexpr bool PE = XF16 && NS == 1;                          // the only stash format the encoding is recomputed for
    DPROF_T(t_pe0);
    if constexpr (NV2 > 0 && PE) {
        if (st.xd != nullptr) {       // chunk 0 before the first tile is issued (once per job: the only drain of the scheme)
            pe_chunk(0);
            grp_wait<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    DPROF_ADD(2, t_pe0);
    auto issue = [&](int slot) {
        const unsigned base = lds0 + (unsigned)slot * GRP_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (st.on[i]) dma16s_stream(src, st.voff[i], __builtin_amdgcn_readfirstlane(base + st.dst[i]));
        src += st.stride;
        if constexpr (NV2 > 0) {
            if (PE && st.xd != nullptr) {
                // re-encode this tile's X2 rows from the quadruples in LDS (chunk ti / 16, loaded a chunk ahead): every wave, 4 rows each
                const f32x4 c = *reinterpret_cast<const f32x4*>(coords + ((ti >> 4) & 1) * GRP_PE_BYTES + (ti & 15) * (KT_ * 16) + st.pe_row16);
                dw_pe_write4(const_cast<char*>(tiles) + slot * GRP_STAGE + st.pe_dst, c, st.pe_gch, lane & 1, st.pe_nvalid);
            } else if (x2_wave) {
                if (st.on2) dma16s_stream(src2, st.voff2, __builtin_amdgcn_readfirstlane(base + st.dst2));
                src2 += st.stride2;
            }
            if (PE && st.xd != nullptr) {      // every wave: its 64 points of the NEXT chunk when a chunk begins
                if ((ti & 15) == 0) pe_chunk(ti / 16 + 1);
                ++ti;
            }
        }
    };
    auto compute = [&](int slot) {
        const char* zt = tiles + slot * GRP_STAGE;
        const char* xt = zt + DMA_OPER;
        const char* x2t = zt + 2 * DMA_OPER;
        // NS = 1: two k-steps of 16 points; NS = 2: one k-step, plane p of the 16 points at rows 16p..16p+15
#pragma unroll
        for (int ks = 0; ks < (NS == 1 ? DMA_KT / 16 : 1); ++ks) {
            if (wave_live) {
                bf16x8 a[NS][4], b[NS][2];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[pl][u] = tr_read<DMA_ROWB>(zt + a_off[u] + (ks + pl) * 16 * DMA_ROWB);
#pragma unroll
                    for (int v = 0; v < 2; ++v) b[pl][v] = xcv(tr_read<DMA_ROWB>(xt + b_off[v] + (ks + pl) * 16 * DMA_ROWB));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        if constexpr (NS == 2) {
                            acc[u][v] = mm(a[1][u], b[0][v], acc[u][v]);
                            acc[u][v] = mm(a[0][u], b[1][v], acc[u][v]);
                        }
                        acc[u][v] = mm(a[0][u], b[0][v], acc[u][v]);
                    }
            }
            if (row_live) {      // (re-reads this wave's row block of Z: 2 LDS reads against keeping a copy of a[wi] alive)
                bf16x8 sel[NS];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) sel[pl] = tr_read<DMA_ROWB>(zt + sel_off + (ks + pl) * 16 * DMA_ROWB);
                if constexpr (NV2 == 0) {
#pragma unroll
                    for (int pl = 0; pl < NS; ++pl) accs[0] = mm(sel[pl], ones, accs[0]);
                } else {
#pragma unroll
                    for (int v = 0; v < NV2; ++v) {
                        bf16x8 bb[NS];
#pragma unroll
                        for (int pl = 0; pl < NS; ++pl) bb[pl] = xcv(tr_read<GRP_X2_ROWB>(x2t + x2_off[v] + (ks + pl) * 16 * GRP_X2_ROWB));
                        // the bias rides in the block's last (padding) column: ones there in plane 0, the padding's zeros in plane 1
                        if (v == NV2 - 1 && bias_lane) bb[0] = ones;
                        if constexpr (NS == 2) {
                            accs[v] = mm(sel[1], bb[0], accs[v]);
                            accs[v] = mm(sel[0], bb[1], accs[v]);
                        }
                        accs[v] = mm(sel[0], bb[0], accs[v]);
                    }
                }
            }
        }
    };
    const bool five = NV2 > 0 && x2_wave && (!(XF16 && NS == 1) || st.xd == nullptr);      // DMAs this wave issues per stage: 4, or 5 with an X2 piece (a
                                                                    // re-encoded X2 is none; the chunk DMAs are not counted: an undercount is safe)
    auto wait_younger = [&](int younger) {          // all but the DMAs of the `younger` most recent stages have landed
        if (younger >= DMA_STAGES - 2) { if (five) grp_wait<5 * (DMA_STAGES - 2)>(); else grp_wait<4 * (DMA_STAGES - 2)>(); }
        else if (younger == 2) { if (five) grp_wait<10>(); else grp_wait<8>(); }
        else if (younger == 1) { if (five) grp_wait<5>(); else grp_wait<4>(); }
        else grp_wait<0>();
    };
    DPROF_T(t_pro);
    for (int t = 0; t < DMA_STAGES - 1 && t < n_tiles; ++t) issue(t);
    int slot = 0;
    const int n_steady = n_tiles - (DMA_STAGES - 1);
#ifdef LUSH_PROF_DW
    bool first = true;
#endif
    DPROF_T(t_steady0);
    for (int t = 0; t < n_steady; ++t) {            // DMA_STAGES - 2 younger stages in flight behind the one awaited
        if (five) grp_wait<5 * (DMA_STAGES - 2)>(); else grp_wait<4 * (DMA_STAGES - 2)>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef LUSH_PROF_DW
        if (first) { DPROF_ADD(3, t_pro); first = false; }
#endif
        issue(slot == 0 ? DMA_STAGES - 1 : slot - 1);
        compute(slot);
        slot = slot + 1 == DMA_STAGES ? 0 : slot + 1;
    }
    DPROF_ADD(4, t_steady0);
    DPROF_T(t_drain);
    for (int t = n_steady < 0 ? 0 : n_steady; t < n_tiles; ++t) {     // drain: nothing left to issue
        wait_younger(n_tiles - 1 - t);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        compute(slot);
        slot = slot + 1 == DMA_STAGES ? 0 : slot + 1;
    }
    DPROF_ADD(5, t_drain);
}

template <bool XF16, bool ZF16, int NS>
__global__ __launch_bounds__(DW_THREADS2) void dw_group_kernel(const DwGroup G) {
    static_assert(DMA_STAGES >= 2 && DMA_STAGES <= 4, "the drain's wait table covers up to 4 stages");
    static_assert(NS == 1 || (NS == 2 && !XF16 && !ZF16), "two planes are bf16 planes");
    constexpr int KT = DMA_KT / NS;                 // points per stage
    extern __shared__ __attribute__((aligned(16))) char tiles[];   // [4 stages][GRP_STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = w >> 2, wi = w & 3;
    DPROF_T(t_kernel);
    LUSH_CLOCK_STAMP(lush_clock_dw, 0);
    // per_job == 2: a flat grid; this workgroup's job and slice from the prefix sums of the jobs' slice counts (a flat grid of
    // exactly as many workgroups as CUs puts 32 of them on each XCD under the dispatcher's round-robin: as a (slices, jobs)
    // grid with idle surplus workgroups some XCDs received 35 working ones, and the launch waited for their second round)
    int my_job = (int)blockIdx.y, my_slice = (int)blockIdx.x;
    if (G.per_job == 2) {
        my_job = 0;
        for (int j = 1; j < G.n; ++j) my_job = (int)blockIdx.x >= G.first[j] ? j : my_job;
        my_slice = (int)blockIdx.x - G.first[my_job];
    }
    // a live-point launch (DwGroup::live_cnt) reads its point count on the device and slices it as the host would have
    int Ppad = G.Ppad, pps = G.per_job == 2 ? G.j[my_job].pps : G.pts_per_split;
    int per_job = G.per_job;
    if (G.live_cnt != nullptr) {
        const int cnt = __builtin_amdgcn_readfirstlane(*G.live_cnt);
        Ppad = (cnt + 255) / 256 * 256;                  // (the stash arrays of a launch are padded to whole 256-point tiles)
        if (Ppad <= LUSH_DW_PERJOB_MAX_PTS && per_job == 0 && (int)gridDim.x >= G.n) {
            // few live points (a trained scene's empty space is dead): ONE job per workgroup on the 1-D grid, as the host chooses
            // for a small pass it knows the size of (lush_abi.hip) -- workgroup b = (slice b / n, job b mod n), just enough slices
            // to fill the chip once.  The walk's 256 slices x 10 jobs of fp32 atomics (0.6 GB at the chip's 1.3 TB/s: 0.45 ms
            // whatever the point count) become slices x 2.3 MB.
            int sp = (int)gridDim.x / G.n;
            const int most = Ppad / LUSH_DW_PERJOB_MIN_PTS;
            sp = sp > most ? most : sp;
            sp = sp < 1 ? 1 : sp;
            if ((int)blockIdx.x >= sp * G.n) return;
            per_job = 1;
            my_slice = (int)blockIdx.x / G.n;
            my_job = (int)blockIdx.x % G.n;
            pps = ((Ppad + sp - 1) / sp + 31) / 32 * 32;
        } else {
            pps = ((Ppad + (int)gridDim.x - 1) / (int)gridDim.x + 31) / 32 * 32;
        }
    }
    const long long p_begin = (long long)my_slice * pps;
    long long p_end = p_begin + pps;
    if (p_end > Ppad) p_end = Ppad;
    const int n_tiles = (int)((p_end - p_begin) / KT);
    if (n_tiles <= 0) return;
    const float unscale = ZF16 ? G.scale[1] : 1.f;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)tiles;
    const int r = lane & 31, h = lane >> 5;
    // job-independent LDS read offsets of this lane
    unsigned a_off[4], b_off[2], x2_off[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) a_off[u] = tr_off_sw(wo * 128 + u * 32, lane);
#pragma unroll
    for (int v = 0; v < 2; ++v) { b_off[v] = tr_off_sw(wi * 64 + v * 32, lane); x2_off[v] = tr_off_x2(v * 32, lane); }
    const unsigned sel_off = tr_off_sw(wo * 128 + wi * 32, lane);

    // (a small launch -- the 4 096-point noise net, configuration 1 -- has fewer slices than the chip has CUs and spent its time
    // walking 7-19 jobs in series, each a ring refill + a few tiles + 64 K atomics: 69 us for 0.4 GFLOP; one job per workgroup then)
    const int j_begin = per_job ? my_job : 0, j_end = per_job ? my_job + 1 : G.n;
    for (int jj = j_begin; jj < j_end; ++jj) {
        int jsel = per_job ? jj : (int)((blockIdx.x + (unsigned)jj) % (unsigned)G.n);
        jsel = __builtin_amdgcn_readfirstlane(jsel);
        const DwJob A = G.j[jsel];                      // one scalar load of the whole record per job
        const bool has_x2 = A.X2 != nullptr;
        const bool wave_live = (wo * 128 < A.n_out) && (wi * 64 < A.k_in);
        const bool row_live = wo * 128 + wi * 32 < A.n_out;           // this wave's 32 rows of the bias / X2 blocks
        const int nv2 = has_x2 ? (A.k2_in > 32 ? 2 : 1) : 0;
        DPROF_T(t_zero);
        // columns beyond a job's widths are never written by its DMAs: start every job from a zeroed ring
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // everyone has left the previous job's tiles
        for (int i = tid; i < DMA_STAGES * GRP_STAGE / 16; i += DW_THREADS2)
            reinterpret_cast<uint4*>(tiles)[i] = make_uint4(0, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // side accumulators: with X2, accs[v] = its 32-column blocks and the bias rides in the last (unused, zero-padded)
        // column of the last block, whose B fragment is forced to ones; without X2, accs[0] = bias against an all-ones B
        f32x16 acc[4][2], accs[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { acc[a][0][q] = 0.f; acc[a][1][q] = 0.f; }
            accs[0][q] = 0.f; accs[1][q] = 0.f;
        }
        DPROF_ADD(1, t_zero);
        const int bias_blk = nv2 ? nv2 - 1 : 0;                    // block that carries the bias column (its column 31)
        const bool x2_wave = has_x2 && w < 4;          // waves 0..3 move the four 1-KiB pieces of the X2 tile
        GrpStream st;
        {
            const int op = w >> 2;                      // waves 0..3 stream Z, 4..7 stream X
            const int ld = op ? A.ldx : A.ldz, ncols = op ? A.k_in : A.n_out, c0 = op ? A.xcol0 : 0;
            const __bf16* base = op ? A.X : A.Z;
            const int plane = ((w & 3) * 8) / KT;       // the tile rows this wave moves, 8 (w & 3) .. + 7, lie in one plane
            st.src = reinterpret_cast<const char*>(base + plane * (op ? A.x_plane : A.z_plane) + p_begin * ld + c0);
            st.stride = (unsigned)(KT * ld * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rp = (w * 4 + i) & 15;
                const int row = 2 * rp + (lane >> 5);                      // tile row: plane row / KT, point row % KT
                const int gch = (lane & 31) ^ ((row & 3) << 2);            // source chunk that belongs at this position
                st.voff[i] = (unsigned)(((row % KT) * ld + gch * 8) * 2);
                st.dst[i] = (unsigned)(op * DMA_OPER + 2 * rp * DMA_ROWB);
                st.on[i] = gch * 8 < ncols;
#ifdef LUSH_ABL_H0      // timing ablation only (wrong results): the job marked pe_mode 9 (layer 1: X = h_0) streams no X
                if (op && A.pe_mode == 9) st.on[i] = false;
#endif
            }
            st.src2 = nullptr; st.stride2 = 0; st.voff2 = 0; st.dst2 = 0; st.on2 = false;
            st.xd = nullptr; st.pe_last = 0; st.pe_row16 = 0; st.pe_dst = 0; st.pe_gch = 0; st.pe_nvalid = 0;
            if (has_x2 && G.xd != nullptr && A.pe_mode != 0) {      // (all waves: each moves its share of the quadruples)
                st.xd = reinterpret_cast<const char*>(G.xd + p_begin * 8 + (A.pe_mode == 2 ? 4 : 0));
                st.pe_last = (long long)Ppad - 1 - p_begin;
            }
            if (st.xd != nullptr) {      // every wave: tile rows 4 w .. 4 w + 3, 16 lanes per row, 4 columns (8 bytes) per lane
                const int row = 4 * w + (lane >> 4);
                st.pe_gch = ((lane >> 1) & 7) ^ (((row >> 1) & 1) << 2);
                st.pe_nvalid = A.k2_in;
                st.pe_row16 = (unsigned)((row % KT) * 16);
                st.pe_dst = (unsigned)(2 * DMA_OPER + 4 * w * GRP_X2_ROWB + lane * 8);
            } else if (x2_wave) {
                const int row = 8 * w + (lane >> 3);
                const int gch = (lane & 7) ^ (((row >> 1) & 1) << 2);
                st.src2 = reinterpret_cast<const char*>(A.X2 + ((8 * w) / KT) * A.x2_plane + p_begin * A.ldx2 + A.x2col0);
                st.stride2 = (unsigned)(KT * A.ldx2 * 2);
                st.voff2 = (unsigned)(((row % KT) * A.ldx2 + gch * 8) * 2);
                st.dst2 = (unsigned)(2 * DMA_OPER + 8 * w * GRP_X2_ROWB);
                st.on2 = gch * 8 < A.k2_in;
            }
        }
        if (nv2 == 0) grp_stream<XF16, ZF16, 0, NS>(st, tiles, lds0, n_tiles, w, lane, wave_live, row_live, x2_wave, a_off, b_off, sel_off, x2_off, acc, accs);
        else if (nv2 == 1) grp_stream<XF16, ZF16, 1, NS>(st, tiles, lds0, n_tiles, w, lane, wave_live, row_live, x2_wave, a_off, b_off, sel_off, x2_off, acc, accs);
        else grp_stream<XF16, ZF16, 2, NS>(st, tiles, lds0, n_tiles, w, lane, wave_live, row_live, x2_wave, a_off, b_off, sel_off, x2_off, acc, accs);
        DPROF_T(t_flush);
#ifdef LUSH_ABL_NOFLUSH      // timing ablation only (wrong results): the accumulators are kept alive, nothing is added
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[2][0]), "v"(acc[2][1]), "v"(acc[3][0]), "v"(acc[3][1]), "v"(accs[0]), "v"(accs[1]));
        if (false)
#endif
        if (row_live) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (v < nv2 || v == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int o = wo * 128 + wi * 32 + acc_row(q, h);
                        const int i = v * 32 + r;
                        if (o < A.n_out) {
                            if (v < nv2 && i < A.k2_in && o < A.n_out2) atomicAdd(A.dW2 + (long long)o * A.ldw2 + A.wcol2 + i, accs[v][q] * unscale);
                            if (v == bias_blk && r == 31) atomicAdd(A.db + o, accs[v][q] * unscale);
                        }
                    }
                }
            }
        }
#ifdef LUSH_ABL_NOFLUSH
        if (false)
#endif
        if (wave_live) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int o = wo * 128 + u * 32 + acc_row(q, h);
                        const int i = wi * 64 + v * 32 + r;
                        if (o < A.n_out && i < A.k_in) atomicAdd(A.dW + (long long)o * A.ldw + A.wcol0 + i, acc[u][v][q] * unscale);
                    }
        }
        DPROF_ADD(6, t_flush);
#ifdef LUSH_PROF_DW
        if (blockIdx.x == 0 && threadIdx.x == 0) lush_prof_dw[8] += 1;
#endif
    }
    DPROF_ADD(0, t_kernel);
    LUSH_CLOCK_STAMP(lush_clock_dw, 1);
#ifdef LUSH_PROF_DW
    if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0) { lush_prof_dw_span[blockIdx.x] = t_kernel; lush_prof_dw_span[1024 + blockIdx.x] = __builtin_amdgcn_s_memtime(); }
#endif
}

// Loss scale of the fp16 gradient chain: scale = 2^k with max|d_raw| * scale in [8, 16) (gradients grow by at most
// ~2^5 through the heads of the sharpest test networks; fp16 tops out at 2^16), 1 when d_raw is all zero or not
// finite.  One launch: per-block maxima by atomicMax on the bit pattern (non-negative floats order as integers),
// the last block to finish writes {scale, 1/scale} and re-arms the two words it used.
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ draw, long long n, float* __restrict__ out,
                                                        unsigned* __restrict__ work /* [2]: running max bits, finished blocks */,
                                                        float* __restrict__ zero_buf, long long zero_n) {
    // (the weight-gradient pass's small scratch accumulators, zeroed here instead of by a fill launch of their own)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (long long)gridDim.x * blockDim.x) zero_buf[i] = 0.f;
    float m = 0.f;
    // 16 bytes per lane and trip (d_raw is [P][4]: n is a multiple of 4 and the rows are 16-byte aligned); with 4-byte loads the
    // fine pass's 42 MB took 34 us
    const float4* d4 = reinterpret_cast<const float4*>(draw);
#pragma unroll 8
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = d4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = n / 4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(draw[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        bool last = gridDim.x == 1;          // (a single workgroup -- the noise net's 4 096 rays -- needs no work words and no zero-fill)
        if (!last) {
            atomicMax(work, __float_as_uint(m));
            __threadfence();
            last = atomicAdd(work + 1, 1u) == gridDim.x - 1;
        }
        if (last) {
            const float mx = gridDim.x == 1 ? m : __uint_as_float(atomicMax(work, 0u));
            float sc = 1.f;
            if (mx > 0.f && mx < 3.0e38f) {
                int e;
                frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
                sc = ldexpf(1.f, 4 - e);        // mx * sc in [8, 16)
            }
            out[0] = sc;
            out[1] = 1.f / sc;
            work[0] = 0u;
            work[1] = 0u;
            __threadfence();
        }
    }
}

// Heads whose dZ is the fp32 d_raw: rgb_linear (X = hv) and alpha_linear (X = h_{NL-1}).
// Thread (c, rr): 16-byte column chunk c of [hv | h_last], every ROWS-th point; fp32 atomics at the end.
template <int NS, bool XF16>
__global__ __launch_bounds__(DW_THREADS) void head_dw_kernel(const float* __restrict__ draw, long long P,
                                                            const __bf16* __restrict__ hv, long long plane_hv, int HV,
                                                            const __bf16* __restrict__ hl, long long plane_h, int HW,
                                                            int pts_per_block, float* __restrict__ dw_rgb,
                                                            float* __restrict__ db_rgb, float* __restrict__ dw_alpha,
                                                            float* __restrict__ db_alpha, const int* __restrict__ live_cnt) {
    if (live_cnt != nullptr) P = *live_cnt;      // (a live-point launch: the rows behind the list hold whatever an earlier launch left)
    const int nchv = HV / 8, nch = nchv + (dw_alpha ? HW / 8 : 0);
    const int rows = DW_THREADS / nch;
    const int c = threadIdx.x % nch, rr = threadIdx.x / nch;
    const long long p0 = (long long)blockIdx.x * pts_per_block;
    long long p1 = p0 + pts_per_block;
    if (p1 > P) p1 = P;
    const bool is_rgb = c < nchv;
    const __bf16* base = is_rgb ? hv + c * 8 : hl + (c - nchv) * 8;
    const long long plane = is_rgb ? plane_hv : plane_h;
    const int ld = is_rgb ? HV : HW;
    float a0[8], a1[8], a2[8], b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = a2[e] = 0.f;
#pragma unroll 4
    for (long long p = p0 + rr; p < (rr < rows ? p1 : p0); p += rows) {
        const float4 d = *reinterpret_cast<const float4*>(draw + p * 4);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + pl * plane + p * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += elem_to_f32<XF16 ? DT_F16 : DT_BF16>(v[e]);
        }
        if (is_rgb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += d.x * x[e]; a1[e] += d.y * x[e]; a2[e] += d.z * x[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a0[e] += d.w * x[e];
        }
        if (c == 0) { b[0] += d.x; b[1] += d.y; b[2] += d.z; b[3] += d.w; }
    }
    // block-level reduction in LDS, then one global atomic per output element and block
    __shared__ float red[3 * 128 + 256 + 4];
    for (int i = threadIdx.x; i < 3 * 128 + 256 + 4; i += DW_THREADS) red[i] = 0.f;
    __syncthreads();
    if (rr < rows) {
        if (is_rgb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[c * 8 + e], a0[e]);
                atomicAdd(&red[128 + c * 8 + e], a1[e]);
                atomicAdd(&red[256 + c * 8 + e], a2[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&red[384 + (c - nchv) * 8 + e], a0[e]);
        }
        if (c == 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&red[640 + e], b[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * HV; i += DW_THREADS) atomicAdd(dw_rgb + i, red[(i / HV) * 128 + (i % HV)]);
    if (dw_alpha != nullptr)
        for (int i = threadIdx.x; i < HW; i += DW_THREADS) atomicAdd(dw_alpha + i, red[384 + i]);
    if (threadIdx.x < 3) atomicAdd(db_rgb + threadIdx.x, red[640 + threadIdx.x]);
    if (threadIdx.x == 3 && db_alpha != nullptr) atomicAdd(db_alpha, red[643]);
}

}  // namespace lush

// ============================================================================
// host launchers (C++ linkage inside the library; the C ABI is in lush_abi.hip)
// ============================================================================
#include "lush_host.h"
#include <cstdio>
#include <cstdlib>

namespace lush {

size_t mlp_fwd_lds_bytes(int hw, int ns, int mt) {
    return (size_t)ns * mt * hw * 2 + (size_t)ns * mt * PE_ROW * 2 + (size_t)mt * 4;
}
size_t mlp_bwd_lds_bytes(int hw, int ns, int mt, int nthreads) {
    return (size_t)ns * mt * hw * 2 + (size_t)mt * DPE_LD * 4 + (size_t)mt * 16 + (size_t)(nthreads / mt) * mt * 24;
}
// tile of the tiled kernels: 64 points, 8 waves (activation + gamma images of 3 planes fill the LDS)
int mlp_fwd_tile(int) { return 64; }
int mlp_bwd_tile(int) { return 64; }

template <class N, int NS, int MT, int NW, bool HAS_ALPHA, int DT = DT_BF16>
static int launch_fwd_k(const MlpFwdArgs& a, int grid, hipStream_t s) {
    auto k = mlp_fwd_kernel<N, NS, MT, NW, HAS_ALPHA, DT>;
    const size_t lds = mlp_fwd_lds_bytes(N::HW, NS, MT);
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, s, a);
    LUSH_HIP(hipGetLastError());
    return 0;
}
template <class N, int NS, int MT, int NW, bool HAS_ALPHA>
static int launch_bwd_k(const MlpBwdArgs& a, int grid, hipStream_t s) {
    auto k = mlp_bwd_kernel<N, NS, MT, NW, HAS_ALPHA>;
    const size_t lds = mlp_bwd_lds_bytes(N::HW, NS, MT, NW * 64);
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, s, a);
    LUSH_HIP(hipGetLastError());
    return 0;
}

// Only the three-plane mode (fp32-grade test mode) still runs on these tiled kernels; 1, 2 planes and fp16 run on
// the chain kernels (lush_mlp_chain.hip).
int launch_mlp_fwd(int net, int ns, const MlpFwdArgs& a, int grid, hipStream_t s) {
    if (ns != 3) return set_error("launch_mlp_fwd: the tiled kernel serves 3 planes only");
    if (net == 0) return launch_fwd_k<NetNerf, 3, 64, 8, true>(a, grid, s);
    return launch_fwd_k<NetNoise, 3, 64, 8, false>(a, grid, s);
}
int launch_mlp_bwd(int net, int ns, const MlpBwdArgs& a, int grid, hipStream_t s) {
    if (ns != 3) return set_error("launch_mlp_bwd: the tiled kernel serves 3 planes only");
    if (net == 0) return launch_bwd_k<NetNerf, 3, 64, 8, true>(a, grid, s);
    return launch_bwd_k<NetNoise, 3, 64, 8, false>(a, grid, s);
}

// ---- feature-layer gradients from G = dZv^T h_{NL-1} (FeatFactorArgs) ----
// part 0: dW_feat[i][k] += sum_v Wva[v][i] G[v][k]   (thread per (i, k), k fastest: G coalesced, Wva broadcast)
// part 1: dW_views[v][i] += sum_k G[v][k] Wf[i][k] + s[v] b_f[i]   (thread per (v, i))
// part 2: db_feat[i] += sum_v Wva[v][i] s[v];  db_views[v] += s[v]
// part 3 (heads folded into the grouped launch): dW_rgb, db_rgb, dW_alpha, db_alpha = hi-plane row + lo-plane row
// Each output has one owner and the launch is ordered after the grouped dW launch on the same stream: plain adds.
constexpr int FF_VB = 8;        // rows v of dW_views per workgroup of part 1 (two per wave)
constexpr int FF_KC = 128;      // columns k of W_feat staged per pass
__global__ __launch_bounds__(256) void feat_factor_kernel(const FeatFactorArgs a) {
    const int HW = a.HW, HV = a.HV;
    const int nb0 = HW * HW / 256, nb1 = (HW / 64) * (HV / FF_VB);
    if ((int)blockIdx.x < nb0) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        const int i = t / HW, k = t % HW;
        // four independent partial sums, 32 products (64 loads) in flight: as one dependent chain of HV L2 loads the kernel was
        // latency-bound, 38-46 us for 0.1 MFLOP; with 8 in flight 19 us
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int v = 0; v < HV; v += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc4[u] += a.w_views[(long long)(v + u) * a.ldv + i] * a.G[(v + u) * HW + k];
        }
        a.g_w_feat[t] += (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        return;
    }
    if ((int)blockIdx.x < nb0 + nb1) {
        // part 1, dW_views[v][i] += sum_k G[v][k] Wf[i][k] + s[v] b_f[i]: both operands are contiguous in k, so with a thread per
        // (v, i) every lane walked its own 1-KiB row of Wf (64 cache lines per load instruction: 19 us per launch).  A workgroup
        // now takes 64 rows i and FF_VB rows v: Wf[i0..i0+63][k] goes through LDS transposed (coalesced reads along k, pitch 65),
        // lane = i reads T[k][lane] without conflicts, and G[v][k] is the same address for the whole wave.
        __shared__ float T[FF_KC * 65];
        const int b = blockIdx.x - nb0, n_it = HW / 64;
        const int i0 = (b % n_it) * 64;
        const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int v0 = __builtin_amdgcn_readfirstlane((b / n_it) * FF_VB + w * 2);
        const float* __restrict__ g0 = a.G + (long long)v0 * HW;
        const float* __restrict__ g1 = g0 + HW;
        const float* __restrict__ wf = a.w_feat + (long long)i0 * HW;
        float acc0 = 0.f, acc1 = 0.f;
        // this wave's two rows of G, column c in lane c % 64 of register c / 64 (FF_KC / 64 registers per pass): read coalesced
        // once per pass and handed out by v_readlane -- as wave-uniform global loads inside the k loop every batch of 16 columns
        // waited for L2 (16 round trips per workgroup: most of the kernel's 14 us)
        for (int kc = 0; kc < HW; kc += FF_KC) {
            float gr0[FF_KC / 64], gr1[FF_KC / 64];
#pragma unroll
            for (int q = 0; q < FF_KC / 64; ++q) { gr0[q] = g0[kc + q * 64 + lane]; gr1[q] = g1[kc + q * 64 + lane]; }
            if (kc) __syncthreads();                       // the previous pass has been read
            float tmp[64 * FF_KC / 256];
#pragma unroll
            for (int j = 0; j < 64 * FF_KC / 256; ++j) {
                const int idx = j * 256 + threadIdx.x, r = idx / FF_KC, c = idx % FF_KC;
                tmp[j] = wf[(long long)r * HW + kc + c];
            }
#pragma unroll
            for (int j = 0; j < 64 * FF_KC / 256; ++j) {
                const int idx = j * 256 + threadIdx.x, r = idx / FF_KC, c = idx % FF_KC;
                T[c * 65 + r] = tmp[j];
            }
            __syncthreads();
#pragma unroll
            for (int c0 = 0; c0 < FF_KC; c0 += 16) {
                float tv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tv[j] = T[(c0 + j) * 65 + lane];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = c0 + j;
                    const float ga = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gr0[c / 64]), c % 64));
                    const float gb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gr1[c / 64]), c % 64));
                    acc0 += ga * tv[j];
                    acc1 += gb * tv[j];
                }
            }
        }
        const float bf = a.b_feat[i0 + lane];
        a.g_w_views[(long long)v0 * a.ldv + i0 + lane] += acc0 + a.s[v0] * bf;
        a.g_w_views[(long long)(v0 + 1) * a.ldv + i0 + lane] += acc1 + a.s[v0 + 1] * bf;
        return;
    }
    const int n2 = HW + HV, n3 = a.Hd ? 3 * HV + 3 + HW + 1 : 0;
    int t = (blockIdx.x - nb0 - nb1) * 256 + threadIdx.x;
    if (t < n2) {
        if (t < HW) {
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int v = 0; v < HV; v += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc4[u] += a.w_views[(long long)(v + u) * a.ldv + t] * a.s[v + u];
            }
            a.g_b_feat[t] += (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        } else {
            a.g_b_views[t - HW] += a.s[t - HW];
        }
        return;
    }
    t -= n2;
    if (t < n3) {     // heads: hi + lo plane rows
        if (t < 3 * HV) {
            const int c = t / HV, j = t % HV;
            a.g_w_rgb[t] += a.Hd[c * HV + j] + a.Hd[(4 + c) * HV + j];
        } else if (t < 3 * HV + 3) {
            const int c = t - 3 * HV;
            a.g_b_rgb[c] += a.sH[c] + a.sH[4 + c];
        } else if (a.g_w_alpha) {
            const int k = t - 3 * HV - 3;
            if (k < HW) a.g_w_alpha[k] += a.G[(HV + 3) * HW + k] + a.G[(HV + 7) * HW + k];
            else a.g_b_alpha[0] += a.s[HV + 3] + a.s[HV + 7];
        }
    }
}
int launch_feat_factor(const FeatFactorArgs& a, hipStream_t s) {
    if (a.HW % FF_KC != 0 || a.HV % FF_VB != 0 || (a.HW * a.HW) % 256 != 0) return set_error("launch_feat_factor: widths must be multiples of 128 (W) and 8 (W/2)");
    const int tail = a.HW + a.HV + (a.Hd ? 3 * a.HV + 3 + a.HW + 1 : 0);
    const int blocks = a.HW * a.HW / 256 + (a.HW / 64) * (a.HV / FF_VB) + (tail + 255) / 256;
    hipLaunchKernelGGL(feat_factor_kernel, dim3(blocks), dim3(256), 0, s, a);
    LUSH_HIP(hipGetLastError());
    return 0;
}

int launch_pack_f32(int net, int ns, const MlpParams& prm, void* packed, hipStream_t s) {
    if (net == 0) {
        float* dst = reinterpret_cast<float*>(packed) + (size_t)NetNerf::total_entries * ns * 256;
        hipLaunchKernelGGL(pack_f32_kernel<NetNerf>, dim3((NetNerf::f32_total + 255) / 256), dim3(256), 0, s, prm, dst);
    } else {
        float* dst = reinterpret_cast<float*>(packed) + (size_t)NetNoise::total_entries * ns * 256;
        hipLaunchKernelGGL(pack_f32_kernel<NetNoise>, dim3((NetNoise::f32_total + 255) / 256), dim3(256), 0, s, prm, dst);
    }
    LUSH_HIP(hipGetLastError());
    return 0;
}

int launch_pack_plan(const void* plan, int blocks, hipStream_t s, float* zero_buf, long long zero_n) {
    hipLaunchKernelGGL(pack_plan_kernel, dim3(blocks), dim3(64), 0, s, reinterpret_cast<const PlanHeader*>(plan), zero_buf, zero_n);
    LUSH_HIP(hipGetLastError());
    return 0;
}

int launch_pack(int ns, const PackTable& t, int total_blocks, void* dst, hipStream_t s) {
    if (ns == PLANES_F16) hipLaunchKernelGGL((pack_kernel<1, DT_F16>), dim3(total_blocks), dim3(64), 0, s, t, (__bf16*)dst);
    else if (ns == 1) hipLaunchKernelGGL((pack_kernel<1, DT_BF16>), dim3(total_blocks), dim3(64), 0, s, t, (__bf16*)dst);
    else if (ns == 2) hipLaunchKernelGGL((pack_kernel<2, DT_BF16>), dim3(total_blocks), dim3(64), 0, s, t, (__bf16*)dst);
    else if (ns == 3) hipLaunchKernelGGL((pack_kernel<3, DT_BF16>), dim3(total_blocks), dim3(64), 0, s, t, (__bf16*)dst);
    else return set_error("launch_pack: bad planes");
    LUSH_HIP(hipGetLastError());
    return 0;
}

template <int NS, bool XF16 = false>
static int launch_dw_t(const DwArgs& a, int splits, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * NS * DwKt<NS>::v * DW_ROW;
    auto k = dw_gemm_kernel<NS, XF16>;
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(splits), dim3(DW_THREADS2), lds, s, a);
    LUSH_HIP(hipGetLastError());
    return 0;
}
int launch_dw(int ns, const DwArgs& a, int splits, hipStream_t s) {
    if (a.n_out > DW_T || a.k_in > DW_T) return set_error("launch_dw: layer wider than 256");
    if (ns != 3 || a.x_f16 || a.z_f16) return set_error("launch_dw: the per-layer kernel serves 3 bf16 planes only (1 and 2 planes: launch_dw_group)");
    return launch_dw_t<3>(a, splits, s);
}

template <bool XF16, bool ZF16, int NS>
static int launch_dw_group_t(const DwGroup& g, int splits, hipStream_t s) {
    const size_t lds = (size_t)DMA_STAGES * GRP_STAGE + (g.xd ? 2 * GRP_PE_BYTES : 0);
    auto k = dw_group_kernel<XF16, ZF16, NS>;
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(splits, g.per_job == 1 ? g.n : 1), dim3(DW_THREADS2), lds, s, g);
    LUSH_HIP(hipGetLastError());
    return 0;
}
LUSH_CLOCK_EXPORT(lush_debug_clock_dw, lush_clock_dw)
#ifdef LUSH_PROF_DW
}  // namespace lush
extern "C" int lush_debug_prof_dw(unsigned long long* out, int reset) {
    LUSH_HIP(hipDeviceSynchronize());
    LUSH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(lush::lush_prof_dw), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {}; LUSH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(lush::lush_prof_dw), z, sizeof(z))); }
    return 0;
}
extern "C" int lush_debug_prof_dw_span(unsigned long long* out) {
    LUSH_HIP(hipDeviceSynchronize());
    LUSH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(lush::lush_prof_dw_span), sizeof(unsigned long long) * 2048));
    return 0;
}
namespace lush {
#endif
// one launch for all the weight-gradient GEMMs of a network pass (ns = 1: one 16-bit plane per operand; 2: two bf16 planes)
int launch_dw_group(const DwGroup& g, int splits, int ns, bool x_f16, bool z_f16, hipStream_t s) {
    for (int i = 0; i < g.n; ++i) {
        const DwJob& j = g.j[i];
        if (j.n_out > DW_T || j.k_in > DW_T || j.n_out < 1 || (j.k_in < 8 && !(j.k_in == 0 && j.X2))) return set_error("launch_dw_group: layer width out of range");
        if (j.pe_mode && g.xd && (ns != 1 || !x_f16)) return set_error("launch_dw_group: the encoding is recomputed for one fp16 plane only");
        if (j.db == nullptr) return set_error("launch_dw_group: every job carries its bias gradient");
        // the bias rides in column 31 of the last 32-column block of X2, which must therefore be a padding column
        if (j.X2 && (j.k2_in < 8 || j.k2_in > 63 || j.k2_in % 32 == 0)) return set_error("launch_dw_group: second input block must leave its last column free");
    }
    if (g.pts_per_split % DMA_KT != 0 || g.Ppad % DMA_KT != 0) return set_error("launch_dw_group: slices are whole 32-point tiles");
    if (ns == 2) {
        if (x_f16 || z_f16) return set_error("launch_dw_group: two planes are bf16 planes");
        return launch_dw_group_t<false, false, 2>(g, splits, s);
    }
    if (ns != 1) return set_error("launch_dw_group: 1 or 2 planes");
    if (z_f16 && g.scale == nullptr) return set_error("launch_dw_group: the fp16 gradient GEMM needs its loss scale");
    if (z_f16) return x_f16 ? launch_dw_group_t<true, true, 1>(g, splits, s) : launch_dw_group_t<false, true, 1>(g, splits, s);
    return x_f16 ? launch_dw_group_t<true, false, 1>(g, splits, s) : launch_dw_group_t<false, false, 1>(g, splits, s);
}

int launch_grad_scale(const float* draw, long long n, float* scale /* {scale, 1/scale, 2 work words} */, float* zero_buf, long long zero_n,
                      hipStream_t s) {
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    if (n <= 65536) blocks = 1;       // 16 K float4 for 256 threads: ~5 us either way, and one launch instead of fill + launch
    if (blocks > 1) LUSH_HIP(hipMemsetAsync(scale, 0, 16, s));      // the work words (the scratch is the caller's fresh memory)
    hipLaunchKernelGGL(grad_scale_kernel, dim3(blocks), dim3(256), 0, s, draw, n, scale, reinterpret_cast<unsigned*>(scale + 2), zero_buf, zero_n);
    LUSH_HIP(hipGetLastError());
    return 0;
}

int launch_head_dw(int ns, bool x_f16, const float* draw, long long P, const __bf16* hv, long long plane_hv, int HV,
                   const __bf16* hl, long long plane_h, int HW, float* dw_rgb, float* db_rgb, float* dw_alpha,
                   float* db_alpha, hipStream_t s, const int* live_cnt) {
    const int ppb = 1024;     // measured on the fine pass (2.6 M points): 128 -> 661 us, 512 / 1024 -> 503, 2048 -> 617, 8192 -> 1097
    const int blocks = (int)((P + ppb - 1) / ppb);
    if (x_f16) hipLaunchKernelGGL((head_dw_kernel<1, true>), dim3(blocks), dim3(DW_THREADS), 0, s, draw, P, hv, plane_hv, HV, hl, plane_h, HW, ppb, dw_rgb, db_rgb, dw_alpha, db_alpha, live_cnt);
    else if (ns == 1) hipLaunchKernelGGL((head_dw_kernel<1, false>), dim3(blocks), dim3(DW_THREADS), 0, s, draw, P, hv, plane_hv, HV, hl, plane_h, HW, ppb, dw_rgb, db_rgb, dw_alpha, db_alpha, live_cnt);
    else if (ns == 2) hipLaunchKernelGGL((head_dw_kernel<2, false>), dim3(blocks), dim3(DW_THREADS), 0, s, draw, P, hv, plane_hv, HV, hl, plane_h, HW, ppb, dw_rgb, db_rgb, dw_alpha, db_alpha, live_cnt);
    else hipLaunchKernelGGL((head_dw_kernel<3, false>), dim3(blocks), dim3(DW_THREADS), 0, s, draw, P, hv, plane_hv, HV, hl, plane_h, HW, ppb, dw_rgb, db_rgb, dw_alpha, db_alpha, live_cnt);
    LUSH_HIP(hipGetLastError());
    return 0;
}

}  // namespace lush
