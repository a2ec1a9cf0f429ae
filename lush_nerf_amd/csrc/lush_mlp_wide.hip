// lush-march: one-plane forward chain of the 8x256 NeRF MLP with 64 points per wave (gfx950 / MI355X).
//
// Same function as mlp_chain_fwd_half_kernel (lush_mlp_chain.hip; utils/run_lushnerf_helpers.py:334-344, 394-423 and
// NeRFAll.mlpforward, models/lushnerf.py:234-266), other organisation.  What the SQ counters of the 32-points-per-wave
// kernels say (profiles/r02_sq_counters.md): 8 non-MFMA instructions per MFMA, matrix pipe 46-48 % busy per SIMD --
// every weight fragment read from LDS feeds ONE MFMA, and the accumulator -> next-operand conversion (ReLU, 16-bit
// pack, decision bits) runs between the GEMM phases, where only a second workgroup on the CU can hide it.  Here:
//
//   * a wave owns 64 points = TWO 32-column MFMA blocks: every A fragment (weights, from the LDS ring) feeds two
//     MFMAs, so fragment reads, ring DMAs, barriers and slot arithmetic per MFMA halve;
//   * a layer is computed in four QUARTER passes of 64 output rows (2 row blocks x 2 column blocks = 64 accumulator
//     registers) over the whole K, with two accumulator sets in ping-pong: while set X accumulates pass p, the VALU
//     converts set Y (pass p-1) into k-blocks of the NEXT layer's B operand as fillers between X's MFMAs, then
//     re-loads Y with the biases of pass p+1.  The conversion therefore hides behind MFMAs of the same wave: one
//     workgroup per CU, one wave per SIMD, 512 registers (two B-operand buffers of 128 registers in ping-pong by layer);
//   * the weight stream (NetT::fwd4 copy) is uniform: positions of 8 KiB = 4 k-blocks x 2 row blocks, one s_barrier
//     per position ("mid-step", as in the chain kernels), 16 MFMAs per wave between two barriers.
//
// The schedule is written out: after every MFMA a fixed list of fillers (fragment read / DMA, conversion units of the
// pending accumulator set, bias re-loads, stash traffic), pinned with sched_barrier.
#include "lush_mlp_wide.h"
#include "lush_host.h"

#include <cstdio>
#include <cstdlib>
#include <utility>

namespace lush {

// PE image of this kernel: rows of 272 bytes (256 + 16), no XOR swizzle.  A 16-lane group of a ds_read_b128 reads 16 different
// rows at the same chunk: with a pitch of 68 dwords their 4-dword pieces fall on 16 disjoint bank quads (68 r mod 64 = 4 r), so
// the fragment reads are conflict-free AND chunk offsets are immediates of the instruction (one address register per column
// block instead of one per (column block, chunk): the XOR-swizzled addresses were hoisted out of the tile loop and spilled).
constexpr int WD_PE_PITCH = PE_ROW * 2 + 16;
constexpr int WD_PE_PLANE = WD_MT * WD_PE_PITCH;
constexpr int WD_SB = 2;                // stash rows in flight between their LDS read-back and their store
enum { WB_NONE = 0, WB_REG = 1, WB_PEX = 2, WB_PED = 3 };     // B operand of a position
enum { WC_NONE = 0, WC_ACT = 1, WC_ALPHA = 2 };               // what happens to the pending accumulator set

constexpr int WD_WRAP = NetNerf::fwd4_len / 8;      // stream positions per tile
static_assert(NetNerf::fwd4_len % 8 == 0, "the quarter-row stream is whole positions");

LUSH_CLOCK_DECL(lush_clock_wide_fwd)
#ifdef LUSH_PROF   // developer build: cycle counts (s_memtime) of block 0 / wave 0, read back through lush_debug_prof_wide
__device__ unsigned long long lush_prof_wide[16];
#define WPROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WPROF_ADD(slot, t0) cx.prof[slot] += __builtin_amdgcn_s_memtime() - (t0)
#else
#define WPROF_T(var)
#define WPROF_ADD(slot, t0)
#endif

// Per-pass runtime parameters (wave-uniform unless noted)
struct WdRt {
    bool pre_on;                   // the leading gamma(x) position runs (skip layer)
    unsigned clamp;                // packed lower clamp of the conversion: 0 = ReLU, 0x80008000 = identity
    char* mptr[2];                 // uniform: decision words of (column block c, row block 0 of the pending set's layer); + 2 lane
    unsigned lane2;                // per-lane: 2 * lane
    const float* nextbias;         // LDS: biases of the rows the pending set accumulates next (row block 0, +8h applied by the reader)
    char* tile;                    // this wave's 4-KiB LDS transposition tile
    char* srows;                   // stash rows of this wave's first point (uniform), bytes
    unsigned soff[2];              // per-lane: byte offset of (row n of column block c, 8 hh) in a [point][HW] stash array
    unsigned srow_off;             // per-lane: byte offset of (row lane>>3, 16-byte chunk lane&7) in a [point][HW] stash array
};

// ---------------------------------------------------------------------------------------------
// conversion of a pending accumulator set, cut into filler units
// ---------------------------------------------------------------------------------------------
// Block b = rbl * 2 + c (row block major: the k-blocks of the next operand complete in k order).  Per block and
// t = 0, 1 (accumulators 8t .. 8t+7 -> k-block KB0 + 2 rbl + t): four pairs i, each one unit [convert, clamp] and,
// with MASK, one unit [zero flags of the pair, merged into the block's decision word]: the flags are
// v_pk_sub_u16(1, x) with clamp (1 where the clamped 16-bit value is zero), shifted into place by a v_lshl_or_b32 and
// inverted once at the end (bit q = accumulator q: mask_index() layout).  VALU units per block: MASK ? 16 : 8.
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
struct WdConvTmp {
    unsigned o[4];       // the packed pairs of the current (block, t)
    unsigned m;          // zero flags of the current block, merged
    unsigned word[4];    // decision word per block
};

template <int NRQP, int KB0, bool MASK>
struct WdConv {
    static constexpr int NB = 2 * NRQP;
    static constexpr int UPT = MASK ? 8 : 4;          // units per (block, t)
    static constexpr int UPB = 2 * UPT;
    static constexpr int NVU = NB * UPB;

    // unit u of (block, t): MASK: cvt0 cvt1 msk0 cvt2 msk1 cvt3 msk2 msk3 ; else cvt0..3
    template <int K>
    static __device__ __forceinline__ void unit(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], WdConvTmp& t, unsigned clamp) {
        constexpr int b = K / UPB, r = K % UPB, tt = r / UPT, u = r % UPT;
        constexpr int c = b % 2, rbl = b / 2, kb = KB0 + 2 * rbl + tt;
        constexpr int kind = MASK ? ((u == 0 || u == 1 || u == 3 || u == 5) ? 0 : 1) : 0;
        constexpr int i = MASK ? (kind == 0 ? (u == 0 ? 0 : (u == 1 ? 1 : (u == 3 ? 2 : 3))) : (u == 2 ? 0 : (u == 4 ? 1 : (u == 6 ? 2 : 3)))) : u;
        // Each unit is ONE volatile asm statement: pure VALU code is otherwise free to leave its filler slot -- the
        // compiler gathered a whole pass's conversions at the head of the basic block, where nothing hides them.
        if constexpr (kind == 0) {
            unsigned o;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, %3"
                         : "=&v"(o) : "v"(pend[c][rbl][8 * tt + 2 * i]), "v"(pend[c][rbl][8 * tt + 2 * i + 1]), "s"(clamp));
            t.o[i] = o;
            if constexpr (i == 3) {      // the k-block leaves as one 128-bit value (one tuple copy into the operand buffer)
                const u32x4 v = {t.o[0], t.o[1], t.o[2], t.o[3]};
                xout[c][kb] = v;
            }
        } else {
            // zero flags of the pair (1 where the clamped halfword is zero) at bits 0 and 16, shifted to 8 tt + 2 i / + 16 and
            // merged; the last unit folds the high halves in and inverts: bit q = accumulator q positive.
            // (v_dot2_u32_u16 would place both flags in one instruction, but costs ~12 cycles and stalls the matrix pipe:
            // tools/micro/mfma_fillers.hip; v_lshl_or_b32 hides behind the MFMAs like any plain VALU instruction.)
            constexpr int sh = 8 * tt + 2 * i;
            unsigned z;
            if constexpr (tt == 0 && i == 0) {
                asm volatile("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(t.m) : "v"(t.o[i]));
            } else if constexpr (tt == 1 && i == 3) {
                unsigned w;
                asm volatile("v_pk_sub_u16 %1, 1, %3 op_sel_hi:[0,1] clamp\n\tv_lshl_or_b32 %2, %1, %4, %2\n\tv_lshrrev_b32 %1, 15, %2\n\tv_or_b32 %1, %1, %2\n\tv_not_b32 %0, %1"
                             : "=v"(w), "=&v"(z), "+v"(t.m) : "v"(t.o[i]), "n"(sh));
                t.word[b] = w;
            } else {
                asm volatile("v_pk_sub_u16 %0, 1, %2 op_sel_hi:[0,1] clamp\n\tv_lshl_or_b32 %1, %0, %3, %1" : "=&v"(z), "+v"(t.m) : "v"(t.o[i]), "n"(sh));
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// one quarter pass
// ---------------------------------------------------------------------------------------------
// MFMAs: [PRE position] + NPOS main positions; a position = 8 units x 2 column blocks, two halves around the
// mid-step.  NRQ = 2: unit u = (k-block u/2, row block u%2); NRQ = 1: unit u = k-block u, row block 0.
//
// One wave per SIMD: an MFMA occupies the matrix pipe for 32 cycles and the wave issues in order, so whatever stands
// between two MFMAs beyond ~6 issue slots (24 cycles) delays the next MFMA and the pipe idles -- and a gap with fewer
// fillers cannot make up for it.  The fillers are therefore laid out by a small compile-time list scheduler: every
// main-position gap g = 16 I + m starts with its FIXED load
//   m 0..3  : fragment read of the position's second half        m 4, 5 : the scalar slot arithmetic of the coming refill
//   m 8     : the refill DMA pair + a fragment of the next position   m 9..11 : fragments of the next position
//   stash   : (layers that stash their input) LDS writes / read-backs in m 12..15, each row stored two gaps behind its read-back
// and the items of the pending accumulator set -- per block its conversion units, its decision-word store and its four
// bias re-loads, in dependency order -- are poured into the gaps up to a cap of issue slots per gap (raised only if the
// pass's deadline D could not be met otherwise).

template <int NRQ, int NPOS, int BMAIN, int KB0, int PRE, int CK, int NRQP, int CKB0, bool MASK, int NRQN, int D, int PQ, bool STASH, int LD>
struct WdPass {
    static constexpr int KBPP = 8 / NRQ;
    static constexpr int NG = NPOS * 16;
    static constexpr int DG = D < NG ? D : NG;
    using CV = WdConv<NRQP, CKB0, MASK>;
    static constexpr int NBLK = CK == WC_ACT ? CV::NB : 0;
    static constexpr int IPB = CK == WC_ACT ? CV::UPB + 5 : 0;      // items per block: VALU units, word store, 4 bias loads
    static constexpr int NIT = NBLK * IPB;
    static_assert(BMAIN == WB_REG || NPOS == 1, "a PE-image segment is one position");
    static_assert(!STASH || NPOS == 4, "the stash pipeline is laid out over the 16 positions of a layer");

    struct Regs {
        bf16x8 a1[4];
        u32x4 bpe[2][4];
        WdConvTmp ct;
        unsigned dma_dst, dma_off;
    };

    // ---- stash pipeline over the layer's 16 positions P = 4 PQ + I.  Job J = (column block J / 4, k-blocks 4 (J % 4) ..)
    // = 32 rows x 128 bytes: LDS writes in position J (gaps 12..15, tile J % 2), read-backs in position J + 1 (gaps 12..15),
    // each row stored two gaps behind its read-back.  The four 128-byte pieces of a 512-byte row are four consecutive jobs, so
    // they reach memory within a few microseconds of each other: with the pieces of a row spread over the whole layer
    // (round-3 first version) the stash-writing forward took 6 % longer -- the memory side merges what arrives together.
    static constexpr int ST_JOBS = 8;
    static constexpr bool st_write(int g) { return STASH && g % 16 >= 12 && (4 * PQ + g / 16) < ST_JOBS; }
    static constexpr bool st_read(int g) { return STASH && g % 16 >= 12 && (4 * PQ + g / 16) >= 1 && (4 * PQ + g / 16) <= ST_JOBS; }
    // row i of job J is stored two gaps behind its read-back: gaps 14, 15 of position J + 1 and 0, 1 of position J + 2 (two rows in
    // flight instead of four; with the stores a position later, in gaps 4..7, the stash-writing forward took 4 % longer)
    static constexpr int st_store_job(int g) {       // job whose row leaves in gap g, or -1
        const int P = 4 * PQ + g / 16, m = g % 16;
        const int J = m >= 14 ? P - 1 : (m < 2 ? P - 2 : -1);
        return (STASH && J >= 0 && J < ST_JOBS) ? J : -1;
    }
    static constexpr int st_store_row(int g) { return g % 16 >= 14 ? g % 16 - 14 : g % 16 + 2; }
    static constexpr bool st_store(int g) { return st_store_job(g) >= 0; }

    // ---- the list scheduler ----
    static constexpr int item_weight(int k) {      // issue slots of item k
        const int r = k % (IPB > 0 ? IPB : 1), b = k / (IPB > 0 ? IPB : 1);
        if (r < CV::UPB) {
            if (!MASK) return 3;
            const int u = r % CV::UPT;
            return (u == 0 || u == 1 || u == 3 || u == 5) ? 3 : 2;       // [cvt, clamp, (accvgpr_write)] / [flags, dot2]
        }
        if (r == CV::UPB) return MASK ? 2 : 0;                              // decision-word store
        return (b / 2) < NRQN ? 1 : 0;                                      // bias re-load (only the row blocks the next pass uses)
    }
    static constexpr int fixed_load(int g) {
        const int m = g % 16;
        int w = 0;
        if (m < 4) w += 1;
        if (m == 4 || m == 5) w += 3;
        if (m == 8) w += 10;
        if (m >= 9 && m <= 11) w += 1;
        if (st_write(g)) w += 1;
        if (st_read(g)) w += 1;
        if (st_store(g)) w += 2;
        return w;
    }
    struct Sched {
        int first[NG + 1];       // items first[g] .. first[g+1]-1 go to gap g
        int cap;
    };
    static constexpr Sched make_sched() {
        Sched S{};
        for (int cap = 6; cap < 64; ++cap) {
            int k = 0;
            for (int g = 0; g < NG; ++g) {
                S.first[g] = k;
                int load = fixed_load(g);
                if (g < DG || g == NG - 1) {
                    while (k < NIT) {
                        const int w = item_weight(k);
                        if (w > 0 && load + w > cap && !(g == NG - 1)) break;
                        load += w;
                        ++k;
                    }
                }
            }
            S.first[NG] = k;
            S.cap = cap;
            // accept when everything before the deadline gap fitted without the overflow into the last gap
            int last = 0;
            for (int g = 0; g < NG; ++g)
                if (S.first[g + 1] > S.first[g]) last = g;
            if (k == NIT && (last < DG || NIT == 0)) return S;
        }
        return S;
    }
    static constexpr Sched SC = make_sched();

    template <int SRC, int NKB>
    static __device__ __forceinline__ void load_bpe(Regs& r, const char* peimg, int row0, int lane) {
        const int n = lane & 31, hh = lane >> 5;
        const char* p = peimg + (row0 + n) * WD_PE_PITCH + hh * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < NKB; ++k)
                r.bpe[c][k] = *reinterpret_cast<const u32x4*>(p + c * 32 * WD_PE_PITCH + ((SRC == WB_PED ? PE_X / 8 : 0) + 2 * k) * 16);
    }

    template <int SRC, int NRQ_, int I, int M>
    static __device__ __forceinline__ void mfma(f32x16 (&act)[2][2], const Regs& r, const WdCarry& cr, const u32x4 (&xin)[2][16]) {
        constexpr int u = M / 2, c = M % 2;
        constexpr int kbl = NRQ_ == 2 ? u / 2 : u, rbl = NRQ_ == 2 ? u % 2 : 0;
        const bf16x8 a = u < 4 ? cr.a0[u] : r.a1[u - 4];
        if constexpr (SRC == WB_REG) act[c][rbl] = mfma_f16(a, __builtin_bit_cast(bf16x8, xin[c][KB0 + I * (8 / NRQ_) + kbl]), act[c][rbl]);
        else act[c][rbl] = mfma_f16(a, __builtin_bit_cast(bf16x8, r.bpe[c][SRC == WB_PED ? kbl % 2 : kbl]), act[c][rbl]);
    }

    // fixed fillers of gap M of a position: fragment reads, the refill's scalar arithmetic, the DMA pair
    template <int M>
    static __device__ __forceinline__ void fixed(WdCtx& cx, Regs& r, WdCarry& cr, const char* rd, const char* rd_next) {
        if constexpr (M < 4) {
            r.a1[M] = *reinterpret_cast<const bf16x8*>(rd + (4 + M) * 1024 + cx.lane * 16);
        } else if constexpr (M == 4) {
            r.dma_dst = cx.dma_base + cx.slot_off;              // refill the slot this position frees ...
            r.dma_off = cx.fetch_off;                           // ... with stream position +S
        } else if constexpr (M == 5) {
            cx.fetch_off = cx.fetch_off + WD_SLOT == (unsigned)WD_WRAP * WD_SLOT ? 0u : cx.fetch_off + WD_SLOT;
            cx.slot_off = cx.slot_off + WD_SLOT == (unsigned)WD_S * WD_SLOT ? 0u : cx.slot_off + WD_SLOT;
        } else if constexpr (M == 8) {
#ifndef LUSH_ABL_NODMA
            wd_dma_pair(cx.gbase + r.dma_off, cx.gbase + r.dma_off + 4096u, cx.voff, r.dma_dst, r.dma_dst + 4096u);
#endif
            cr.a0[0] = *reinterpret_cast<const bf16x8*>(rd_next + cx.lane * 16);
        } else if constexpr (M >= 9 && M <= 11) {
            cr.a0[M - 8] = *reinterpret_cast<const bf16x8*>(rd_next + (M - 8) * 1024 + cx.lane * 16);
        }
    }

    // item K of the pending set's stream
    template <int K>
    static __device__ __forceinline__ void item(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], Regs& r, const WdRt& rt, int lane) {
        constexpr int b = K / IPB, rr = K % IPB, c = b % 2, rbl = b / 2;
        if constexpr (rr < CV::UPB) {
            CV::template unit<b * CV::UPB + rr>(pend, xout, r.ct, rt.clamp);
        } else if constexpr (rr == CV::UPB) {
            if constexpr (MASK) *reinterpret_cast<unsigned short*>(rt.mptr[c] + rbl * 128 + rt.lane2) = (unsigned short)r.ct.word[b];
        } else if constexpr (rbl < NRQN) {
            constexpr int j = rr - CV::UPB;       // 1..4
            const f32x4 v = *reinterpret_cast<const f32x4*>(rt.nextbias + rbl * 32 + 8 * (lane >> 5) + (j == 1 ? 0 : j == 2 ? 4 : j == 3 ? 16 : 20));
#pragma unroll
            for (int e = 0; e < 4; ++e) pend[c][rbl][4 * (j - 1) + e] = v[e];
        }
    }

    template <int G>
    static __device__ __forceinline__ void pending(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], Regs& r, const WdRt& rt, int lane, float (&alpha)[2]) {
#ifdef LUSH_ABL_NOCONV      // timing ablation only (wrong results): the MFMA + fragment / DMA skeleton alone
        return;
#endif
        if constexpr (CK == WC_ACT) {
            wd_unroll<SC.first[G], SC.first[G + 1]>([&](auto kc) __attribute__((always_inline)) { item<decltype(kc)::value>(pend, xout, r, rt, lane); });
        } else if constexpr (CK == WC_ALPHA) {
            if constexpr (G == 0) {
                alpha[0] = pend[0][0][0];
                alpha[1] = pend[1][0][0];
            }
            if constexpr (G == 1) {
                // Only row 0 of the alpha head's two accumulator blocks is ever read, so for the register allocator the other 15
                // registers of each block die with the block's LAST MFMA -- and it handed them to the outputs of the conversion
                // units' asm statements in the very next gaps, 2 .. 9 wait states behind an MFMA that writes all 16 for 11
                // (isa_check.py rule R4; hipcc pads its own instructions there, not an asm statement's).  Whole blocks stay
                // live until here, two MFMAs (16 wait states) later.
                asm volatile("" ::"v"(pend[0][0]), "v"(pend[1][0]));
            }
            if constexpr (G >= 1 && G <= 8) {       // re-load both blocks of each column set with the next pass's biases
                constexpr int q = G - 1, c = q % 2, rbl = (q / 2) % 2, jj = q / 4;    // two loads per gap
                if constexpr (rbl < NRQN) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int j = 1 + 2 * jj + j2;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(rt.nextbias + rbl * 32 + 8 * (lane >> 5) + (j == 1 ? 0 : j == 2 ? 4 : j == 3 ? 16 : 20));
#pragma unroll
                        for (int e = 0; e < 4; ++e) pend[c][rbl][4 * (j - 1) + e] = v[e];
                    }
                }
            }
        }
    }

    template <int JOB, int I4>
    static __device__ __forceinline__ void stash_store(const WdCarry& cr, const WdRt& rt) {
        constexpr int c = JOB / 4, j = JOB % 4;      // uniform row-block base + ONE per-lane offset
#ifndef LUSH_ABL_NOSTORE
#ifdef LUSH_PLAIN_STASH      // developer A/B: cached stores (tools/micro/mall_probe.hip: a pure write stream is faster with them)
        *reinterpret_cast<u32x4*>(rt.srows + ((c * 32 + 8 * I4) * LD + j * 64) * 2 + rt.srow_off) = cr.sb[I4 % WD_SB];
#else
        __builtin_nontemporal_store(cr.sb[I4 % WD_SB], reinterpret_cast<u32x4*>(rt.srows + ((c * 32 + 8 * I4) * LD + j * 64) * 2 + rt.srow_off));
#endif
#else
        asm volatile("" ::"v"(cr.sb[I4 % WD_SB]));
#endif
    }
    template <int G>
    static __device__ __forceinline__ void stash(const u32x4 (&xin)[2][16], WdCarry& cr, const WdRt& rt, int lane) {
#ifdef LUSH_ABL_NOSTASH      // timing ablation only (wrong results): no LDS transposition, no row stores
        return;
#endif
        constexpr int P = 4 * PQ + G / 16, m = G % 16;
        if constexpr (st_store(G)) stash_store<st_store_job(G), st_store_row(G)>(cr, rt);
        if constexpr (st_write(G)) {
            constexpr int job = P, c = job / 4, j = job % 4, o = m - 12;
            const int n = lane & 31, hh = lane >> 5;
            *reinterpret_cast<u32x4*>(rt.tile + (job % 2) * 4096 + n * 128 + (((2 * o + hh) ^ (n & 7)) << 4)) = xin[c][4 * j + o];
        }
        if constexpr (st_read(G)) {
            constexpr int job = P - 1, i = m - 12;
            const int row = 8 * i + (lane >> 3);
            cr.sb[i % WD_SB] = *reinterpret_cast<const u32x4*>(rt.tile + (job % 2) * 4096 + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        }
    }

    // stores this wave issues in main gaps [g0, g1) (vmcnt bookkeeping; only what is certain: an undercount is safe)
    static constexpr int stores_in(int g0, int g1) {
        int n = 0;
        for (int g = g0 < 0 ? 0 : g0; g < g1 && g < NG; ++g) {
            if (st_store(g)) ++n;
            if (CK == WC_ACT && MASK)
                for (int k = SC.first[g]; k < SC.first[g + 1]; ++k)
                    if (k % IPB == CV::UPB) ++n;
        }
        return n;
    }

    template <int SRC, int NRQ_, int I, bool MAIN>
    static __device__ __forceinline__ void position(WdCtx& cx, f32x16 (&act)[2][2], f32x16 (&pend)[2][2], const u32x4 (&xin)[2][16],
                                                    u32x4 (&xout)[2][16], Regs& r, WdCarry& cr, const char* peimg, int row0, const WdRt& rt,
                                                    float (&alpha)[2]) {
        // (gamma(d) is 2 k-blocks; the stream pads its K to 4 with zero weights: those MFMAs re-use the two real operands)
        if constexpr (SRC != WB_REG) load_bpe<SRC, (SRC == WB_PED ? 2 : 4)>(r, peimg, row0, cx.lane);
        const char* rd = cx.ring + cx.slot_off;
        __builtin_amdgcn_sched_barrier(0);
        wd_unroll<0, 8>([&](auto mc) __attribute__((always_inline)) {
            constexpr int M = decltype(mc)::value;
            mfma<SRC, NRQ_, I, M>(act, r, cr, xin);
            fixed<M>(cx, r, cr, rd, nullptr);
            if constexpr (MAIN) {
                stash<16 * I + M>(xin, cr, rt, cx.lane);
                pending<16 * I + M>(pend, xout, r, rt, cx.lane, alpha);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // mid-step: my pieces of position +1 have landed (the DMAs of +2 .. +S-1 and the stores issued since are younger);
        // after the barrier everyone's have, and nobody reads this position's slot any more
        constexpr int younger = 2 * (WD_S - 2) + (MAIN ? stores_in(16 * (I + 1 - WD_S) + 8, 16 * I + 8) : 0);
        WPROF_T(t_w0);
#ifndef LUSH_ABL_NOVMWAIT
        wd_wait_vm<(younger < 63 ? younger : 63)>();
#endif
        WPROF_T(t_w1);
#ifndef LUSH_ABL_NOBAR
        lds_barrier();
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef LUSH_PROF
        cx.prof[5] += t_w1 - t_w0;
        cx.prof[6] += __builtin_amdgcn_s_memtime() - t_w1;
        cx.prof[7] += 1;
#endif
        const char* rd_next = cx.ring + cx.slot_off;        // (slot_off already names the next slot: gap 5)
        __builtin_amdgcn_sched_barrier(0);
        wd_unroll<8, 16>([&](auto mc) __attribute__((always_inline)) {
            constexpr int M = decltype(mc)::value;
            mfma<SRC, NRQ_, I, M>(act, r, cr, xin);
            fixed<M>(cx, r, cr, rd, rd_next);
            if constexpr (MAIN) {
                stash<16 * I + M>(xin, cr, rt, cx.lane);
                pending<16 * I + M>(pend, xout, r, rt, cx.lane, alpha);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    static __device__ __forceinline__ void run(WdCtx& cx, f32x16 (&act)[2][2], f32x16 (&pend)[2][2], const u32x4 (&xin)[2][16],
                                               u32x4 (&xout)[2][16], WdCarry& cr, const char* peimg, int row0, const WdRt& rt, float (&alpha)[2]) {
        Regs r;
        if constexpr (PRE == WB_PEX) {
            if (rt.pre_on) position<WB_PEX, 2, 0, false>(cx, act, pend, xin, xout, r, cr, peimg, row0, rt, alpha);
        } else if constexpr (PRE == WB_PED) {
            position<WB_PED, 2, 0, false>(cx, act, pend, xin, xout, r, cr, peimg, row0, rt, alpha);
        }
        wd_unroll<0, NPOS>([&](auto ic) __attribute__((always_inline)) {
            position<BMAIN, NRQ, decltype(ic)::value, true>(cx, act, pend, xin, xout, r, cr, peimg, row0, rt, alpha);
        });

    }
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// gamma(x), gamma(d) of the tile's 256 points as fp16 into the padded image, one point per thread
// (utils/run_lushnerf_helpers.py:334-361).  With ONE workgroup per CU nothing hides this prologue, so it is built for
// speed: hardware sin / cos behind an exact range reduction (sincos_rev, lush_mlp_dev.h: 4.2e-7 absolute, 1/500 of the
// fp16 grid this kernel rounds the result to), and the row leaves as twelve 16-byte LDS writes.
__device__ __noinline__ void wd_pe_tile(char* peimg, const float* rays, const float* z, int S, int P, long long tile_pt0, int tid, float* xd,
                                        const int* live) {
    const long long gpt = tile_pt0 + tid;
    float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (gpt < P) point_of(rays, z, S, live ? (long long)live[gpt] : gpt, x, d);      // (a live-point launch: its i-th point is grid point live[i])
    if (xd != nullptr) {      // what the weight gradients re-encode (dw_pe_write, lush_mlp.hip): 32 bytes instead of the 256-byte row
        float4* o = reinterpret_cast<float4*>(xd + gpt * 8);
        o[0] = make_float4(x[0], x[1], x[2], 0.f);
        o[1] = make_float4(d[0], d[1], d[2], 0.f);
    }
    float v[PE_X + PE_D];
#pragma unroll
    for (int c = 0; c < PE_X + PE_D; ++c) v[c] = 0.f;      // (col 63 and cols 91..95: zero padding the K loops do read)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = x[i];
        v[PE_X + i] = d[i];
        float hx, lx, hd, ld;
        rev_split(x[i], &hx, &lx);
        rev_split(d[i], &hd, &ld);
#pragma unroll
        for (int k = 0; k < L_X; ++k) sincos_rev(hx, lx, k, &v[3 + 6 * k + i], &v[3 + 6 * k + 3 + i]);
#pragma unroll
        for (int k = 0; k < L_D; ++k) sincos_rev(hd, ld, k, &v[PE_X + 3 + 6 * k + i], &v[PE_X + 3 + 6 * k + 3 + i]);
    }
    char* row = peimg + tid * WD_PE_PITCH;
#pragma unroll
    for (int c = 0; c < (PE_X + PE_D) / 8; ++c) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned o[1];
            split_pair<1, DT_F16>(v[8 * c + 2 * j], v[8 * c + 2 * j + 1], o);
            w[j] = o[0];
        }
        *reinterpret_cast<u32x4*>(row + c * 16) = w;
    }
}

// acc[c][rbl][q] = b[32 rbl + 16 (q>>3) + 8 h + (q&7)] from the LDS bias block (chain_row() order, as ch_bias)
__device__ __forceinline__ void wd_bias(f32x16 (&acc)[2][2], const float* b, int h, int nrq) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int rbl = 0; rbl < 2; ++rbl) {
            if (rbl < nrq) {
                const f32x4* p = reinterpret_cast<const f32x4*>(b + rbl * 32 + 8 * h);
                const f32x4 v0 = p[0], v1 = p[1], v2 = p[4], v3 = p[5];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[c][rbl][e] = v0[e];
                    acc[c][rbl][4 + e] = v1[e];
                    acc[c][rbl][8 + e] = v2[e];
                    acc[c][rbl][12 + e] = v3[e];
                }
            }
        }
}

template <class N, int SPK>
__global__ __launch_bounds__(WD_NT) void mlp_wide_fwd_kernel(const MlpFwdArgs A) {
    static_assert(N::HW == 256 && N::NL == 8 && N::HV == 128, "the wide kernel is built for the 8x256 net");
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, NRB = N::NRB;
    constexpr int NBIAS = N::f32_w_rgb;
    constexpr bool MASK = SPK > 0;
    constexpr int ML_BYTES = NRB * 128;                        // decision words of one (column block, layer)
    constexpr int CB_BYTES = N::n_mask_layers * ML_BYTES;      // ... of one column block
    LUSH_CLOCK_STAMP(lush_clock_wide_fwd, 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The bias block comes first: its reads then are one per-lane register + an immediate (DS offsets reach 64 KiB; placed
    // behind the images every distinct bias address became a register of its own, hoisted out of the tile loop and spilled).
    float* biasl = reinterpret_cast<float*>(smem);             // [NBIAS] fp32
    char* ring = smem + NBIAS * 4;                             // [WD_S][8 KiB]
    char* stage = ring + WD_S * WD_SLOT;                       // [4 waves][2][4 KiB] stash transposition tiles (SPK > 0)
    char* peimg = stage + 8 * 4096;                            // [256 points][272 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const char* wbase = reinterpret_cast<const char*>(A.wpk);
    {
        const float* f32 = reinterpret_cast<const float*>(wbase + (long long)N::total_entries * 1024);
        for (int i = tid; i < NBIAS; i += WD_NT) biasl[i] = f32[i];
    }
    WdCtx cx;
    cx.ring = ring;
    cx.ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    cx.gbase = wbase + (long long)N::fwd4_base * 1024;
    cx.slot_off = 0;
    cx.fetch_off = (unsigned)(WD_S % WD_WRAP) * WD_SLOT;
    cx.dma_base = __builtin_amdgcn_readfirstlane(cx.ring_lds + (unsigned)w * 1024u);
    cx.w = w;
    cx.lane = lane;
    cx.voff = (unsigned)lane * 16u + (unsigned)w * 1024u;
#ifdef LUSH_PROF
    for (int i = 0; i < 16; ++i) cx.prof[i] = 0;
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < WD_S; ++j)
        wd_dma_pair(cx.gbase + (unsigned)j * WD_SLOT, cx.gbase + (unsigned)j * WD_SLOT + 4096u, cx.voff, cx.dma_base + (unsigned)j * WD_SLOT, cx.dma_base + (unsigned)j * WD_SLOT + 4096u);

    const int row0 = w * 64;
    WdRt rt;
    rt.tile = stage + w * 8192;
    rt.pre_on = false;
    rt.srows = nullptr;
#pragma unroll
    for (int c = 0; c < 2; ++c) rt.soff[c] = (unsigned)(((c * 32 + n) * HW + h * 8) * 2);
    rt.srow_off = (unsigned)(((lane >> 3) * HW + (lane & 7) * 8) * 2);
    rt.lane2 = (unsigned)lane * 2u;
    float alpha[2] = {0.f, 0.f};
    // a live-point launch reads its point count on the device (the host sized the grid for all the points)
    int P = A.P, n_tiles = A.n_tiles;
    if (A.live_cnt != nullptr) {
        P = __builtin_amdgcn_readfirstlane(*A.live_cnt);
        n_tiles = (P + WD_MT - 1) / WD_MT;
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long pt0 = (long long)tile * WD_MT;
        const long long wpt = pt0 + row0;
        WPROF_T(t_tile);
#ifndef LUSH_ABL_NOPE
        wd_pe_tile(peimg, A.rays, A.z, A.S, P, pt0, tid, (SPK > 0 && !A.pe_rows) ? A.xd : nullptr, A.live_idx);
#endif
        // my pieces of the tile's first position have landed.  First tile: the prologue issued S positions, S-1 are younger.
        // Later tiles: that DMA left S positions ago, in the views layer's second quarter; younger are the DMAs of S-1
        // positions and AT LEAST the 16 stash stores of the views hidden + the raw store (an undercount is safe) -- a full
        // drain here would wait for every stash store of the tile to reach memory (microseconds, 40 times per launch).
        if (tile == (int)blockIdx.x) wd_wait_vm<2 * (WD_S - 1)>();
        else wd_wait_vm<2 * (WD_S - 1) + (SPK > 0 ? 17 : 1)>();
        lds_barrier();
        if (SPK > 0 && A.pe_rows) {      // the encoded rows for weight-gradient kernels that read them (variants)
            for (int i = tid; i < WD_MT * 12; i += WD_NT) {
                const int c = i % 12, pt = i / 12;
                const uint4 v = *reinterpret_cast<const uint4*>(peimg + pt * WD_PE_PITCH + c * 16);
                *reinterpret_cast<uint4*>(A.pe + (pt0 + pt) * PE_ROW + c * 8) = v;
            }
        }
        {   // opaque per tile (keeps the static stream addresses from being hoisted out of the tile loop)
            unsigned long long gb = (unsigned long long)cx.gbase;
            asm volatile("" : "+s"(gb));
            cx.gbase = (const char*)gb;
        }
        // decision words of (this wave's column block c, layer ml, row block rb): mbase[c] + ml * ML_BYTES + rb * 128 (+ 2 lane)
        char* mbase[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) mbase[c] = reinterpret_cast<char*>(A.mask) + (wpt / 32 + c) * CB_BYTES;
        char* mdummy = reinterpret_cast<char*>(A.mask_dummy);
        auto mset = [&](int ml, int rb, bool dummy) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 2; ++c) rt.mptr[c] = dummy ? mdummy + c * 256 : mbase[c] + ml * ML_BYTES + rb * 128;
        };

        f32x16 accA[2][2], accB[2][2];
        u32x4 B0[2][16], B1[2][16];
        WdCarry a0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a0.a0[i] = *reinterpret_cast<const bf16x8*>(cx.ring + cx.slot_off + i * 1024 + lane * 16);
        const float* btr = biasl + N::f32_b_trunk;
        wd_bias(accA, btr, h, 2);
        wd_bias(accB, btr + 64, h, 2);

        WPROF_ADD(1, t_tile);
        WPROF_T(t_l0);
        // ---- layer 0: gamma(x) from the PE image, four quarter passes of one position ----
        rt.clamp = 0u;
        WdPass<2, 1, WB_PEX, 0, WB_NONE, WC_NONE, 2, 0, MASK, 2, 16, 0, false, HW>::run(cx, accA, accB, B0, B0, a0, peimg, row0, rt, alpha);
        mset(0, 0, false); rt.nextbias = btr + 128;
        WdPass<2, 1, WB_PEX, 0, WB_NONE, WC_ACT, 2, 0, MASK, 2, 16, 0, false, HW>::run(cx, accB, accA, B0, B0, a0, peimg, row0, rt, alpha);
        mset(0, 2, false); rt.nextbias = btr + 192;
        WdPass<2, 1, WB_PEX, 0, WB_NONE, WC_ACT, 2, 4, MASK, 2, 16, 0, false, HW>::run(cx, accA, accB, B0, B0, a0, peimg, row0, rt, alpha);
        mset(0, 4, false); rt.nextbias = btr + HW;
        WdPass<2, 1, WB_PEX, 0, WB_NONE, WC_ACT, 2, 8, MASK, 2, 16, 0, false, HW>::run(cx, accB, accA, B0, B0, a0, peimg, row0, rt, alpha);

        // ---- layers 1 .. NL-1 and the feature layer (l = NL: no activation, no decision words), two per iteration so
        // that the B-operand buffers swap roles without copies: odd member B0 -> B1, even member B1 -> B0 ----
        auto layer = [&](int l, u32x4 (&xin)[2][16], u32x4 (&xout)[2][16]) __attribute__((always_inline)) {
            const bool feat = l == NL;
            const float* bl = btr + l * HW;                         // (the feature biases follow the trunk biases)
            rt.pre_on = l == N::SKIP;
#ifdef LUSH_ABL_H0      // timing ablation only (wrong results): the h_0 rows go to one tile's worth of rows per workgroup (no HBM traffic)
            rt.srows = reinterpret_cast<char*>(A.h0 + (long long)(l - 1) * A.h_stride + (l == 1 ? (long long)blockIdx.x * WD_MT + (wpt % WD_MT) : wpt) * HW);
#else
            rt.srows = reinterpret_cast<char*>(A.h0 + (long long)(l - 1) * A.h_stride + wpt * HW);
#endif
            // pass 0: set A accumulates rows 0..63; set B (last quarter of the previous layer) -> xin k-blocks 12..15
            rt.clamp = 0u; mset(l - 1, 6, false); rt.nextbias = bl + 64;
            WdPass<2, 4, WB_REG, 0, WB_PEX, WC_ACT, 2, 12, MASK, 2, 48, 0, (SPK > 0), HW>::run(cx, accA, accB, xin, xin, a0, peimg, row0, rt, alpha);
            rt.clamp = feat ? 0x80008000u : 0u;
            mset(l, 0, feat); rt.nextbias = bl + 128;
            WdPass<2, 4, WB_REG, 0, WB_PEX, WC_ACT, 2, 0, MASK, 2, 60, 1, (SPK > 0), HW>::run(cx, accB, accA, xin, xout, a0, peimg, row0, rt, alpha);
            mset(l, 2, feat); rt.nextbias = bl + 192;
            WdPass<2, 4, WB_REG, 0, WB_PEX, WC_ACT, 2, 4, MASK, 2, 60, 2, (SPK > 0), HW>::run(cx, accA, accB, xin, xout, a0, peimg, row0, rt, alpha);
            mset(l, 4, feat); rt.nextbias = feat ? biasl + N::f32_b_alpha : bl + HW;
            WdPass<2, 4, WB_REG, 0, WB_PEX, WC_ACT, 2, 8, MASK, 2, 60, 3, (SPK > 0), HW>::run(cx, accB, accA, xin, xout, a0, peimg, row0, rt, alpha);
        };
        WPROF_ADD(2, t_l0);
        WPROF_T(t_trunk);
#pragma unroll 1
        for (int l2 = 0; l2 < NL / 2; ++l2) {
            layer(1 + 2 * l2, B0, B1);
            layer(2 + 2 * l2, B1, B0);
        }
        WPROF_ADD(3, t_trunk);
        WPROF_T(t_tail);
        // now: B1 = h_{NL-1}, B0 = feature k-blocks 0..11; set B holds the feature layer's last quarter, set A the alpha bias
        rt.pre_on = false;
        // ---- alpha head on h_{NL-1} (1 row block, 2 positions); set B -> feature k-blocks 12..15 ----
        rt.clamp = 0x80008000u; mset(0, 0, true); rt.nextbias = biasl + N::f32_b_views;
        WdPass<1, 2, WB_REG, 0, WB_NONE, WC_ACT, 2, 12, MASK, 2, 32, 0, false, HW>::run(cx, accA, accB, B1, B0, a0, peimg, row0, rt, alpha);
        // ---- views layer: relu(Wv [feature ; gamma(d)] + b), two quarter passes ----
        rt.nextbias = biasl + N::f32_b_views + 64;
        WdPass<2, 4, WB_REG, 0, WB_PED, WC_ALPHA, 1, 0, MASK, 2, 60, 0, false, HW>::run(cx, accB, accA, B0, B0, a0, peimg, row0, rt, alpha);
        rt.clamp = 0u; mset(NL, 0, false); rt.nextbias = biasl + N::f32_b_rgb;
        WdPass<2, 4, WB_REG, 0, WB_PED, WC_ACT, 2, 0, MASK, 1, 60, 0, false, HW>::run(cx, accA, accB, B0, B1, a0, peimg, row0, rt, alpha);
        {   // the second views quarter has no MFMAs left to hide behind
            WdConvTmp ct;
            using CV = WdConv<2, 4, MASK>;
            mset(NL, 2, false);
            wd_unroll<0, CV::NVU>([&](auto kc) __attribute__((always_inline)) { CV::template unit<decltype(kc)::value>(accA, B1, ct, 0u); });
            if constexpr (MASK) {
#pragma unroll
                for (int b = 0; b < 4; ++b) *reinterpret_cast<unsigned short*>(rt.mptr[b % 2] + (b / 2) * 128 + rt.lane2) = (unsigned short)ct.word[b];
            }
        }
        if constexpr (SPK > 0) {    // views hidden -> stash rows [point][HV]
            char* hvrows = reinterpret_cast<char*>(A.hv + wpt * HV);
#ifndef LUSH_WD_DIRECT_STASH
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < N::KKV / 4; ++j) {
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
                        *reinterpret_cast<u32x4*>(rt.tile + n * 128 + (((2 * kq + h) ^ (n & 7)) << 4)) = B1[c][4 * j + kq];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 8 * i + (lane >> 3);
                        const u32x4 v = *reinterpret_cast<const u32x4*>(rt.tile + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(hvrows + ((c * 32 + 8 * i) * HV + j * 64) * 2 + (unsigned)(((lane >> 3) * HV + (lane & 7) * 8) * 2)));
                    }
                }
#else
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned off = (unsigned)(((c * 32 + n) * HV + h * 8) * 2);
#pragma unroll
                for (int kb = 0; kb < N::KKV; ++kb) __builtin_nontemporal_store(B1[c][kb], reinterpret_cast<u32x4*>(hvrows + off + kb * 32));
            }
#endif
        }
        // ---- rgb head (1 row block, 1 position) ----
        WdPass<1, 1, WB_REG, 0, WB_NONE, WC_NONE, 2, 0, MASK, 2, 16, 0, false, HW>::run(cx, accB, accA, B1, B1, a0, peimg, row0, rt, alpha);
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long gpt = wpt + c * 32 + n;
                if (gpt < P && A.live_idx == nullptr) {      // (a live-point launch re-computes points whose raw output the caller has)
                    float4 o;
                    o.x = accB[c][0][0];
                    o.y = accB[c][0][1];
                    o.z = accB[c][0][2];
                    o.w = alpha[c];
                    *reinterpret_cast<float4*>(A.raw + gpt * 4) = o;
                }
            }
        }
        WPROF_ADD(4, t_tail);
#ifdef LUSH_PROF
        cx.prof[8] += 1;
#endif
    }
#ifdef LUSH_PROF
    if (blockIdx.x == 0 && tid == 0) {
        cx.prof[0] = __builtin_amdgcn_s_memtime() - t_kernel;
        for (int i = 0; i < 16; ++i) lush_prof_wide[i] = cx.prof[i];
    }
#endif
    wd_wait_vm<0>();          // the look-ahead DMAs of the non-existent next tile must land before the LDS is released
    LUSH_CLOCK_STAMP(lush_clock_wide_fwd, 1);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
LUSH_CLOCK_EXPORT(lush_debug_clock_fwd, lush_clock_wide_fwd)
static_assert((NetNerf::f32_w_rgb * 4) % 16 == 0, "the ring behind the bias block stays 16-byte aligned");
size_t mlp_wide_fwd_lds_bytes() { return (size_t)WD_S * WD_SLOT + (size_t)WD_PE_PLANE + (size_t)NetNerf::f32_w_rgb * 4 + 8 * 4096; }

template <int SPK>
static int launch_wide_sp(const MlpFwdArgs& a, hipStream_t s) {
    auto k = mlp_wide_fwd_kernel<NetNerf, SPK>;
    const size_t lds = mlp_wide_fwd_lds_bytes();
    static KernelOnce once;              // (per instantiation and device: the attribute call costs more than the launch in the small configs)
    int dev = 0, n_cu = 0;
    if (int rc = current_device_cus(dev, n_cu)) return rc;
    if (int rc = kernel_lds_once(once, dev, reinterpret_cast<const void*>(k), lds)) return rc;
    const int tiles = (a.n_tiles * 128 + WD_MT - 1) / WD_MT;      // a.n_tiles counts 128-point tiles (point arrays are padded to 256)
    MlpFwdArgs b = a;
    b.n_tiles = tiles;
    const int grid = tiles < n_cu ? tiles : n_cu;                 // one workgroup per CU, tiles strided
    hipLaunchKernelGGL(k, dim3(grid), dim3(WD_NT), lds, s, b);
    LUSH_HIP(hipGetLastError());
    return 0;
}

#ifdef LUSH_PROF
extern "C" int lush_debug_prof_wide(unsigned long long* out) {
    LUSH_HIP(hipDeviceSynchronize());
    LUSH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(lush_prof_wide), sizeof(unsigned long long) * 16));
    return 0;
}
#endif

// One fp16 plane, the 8x256 net: a.stash_planes 0 (inference) or 1.
int launch_mlp_wide_fwd(const MlpFwdArgs& a, hipStream_t s) {
    const int sp = a.write_stash ? a.stash_planes : 0;
    if (sp == 0) return launch_wide_sp<0>(a, s);
    if (sp == 1) return launch_wide_sp<1>(a, s);
    return set_error("launch_mlp_wide_fwd: one stash plane at most");
}

}  // namespace lush
