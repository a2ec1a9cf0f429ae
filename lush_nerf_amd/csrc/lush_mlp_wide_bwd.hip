// lush-march: loss-scaled fp16 gradient chain of the 8x256 NeRF MLP with 64 points per wave (gfx950 / MI355X).
//
// Same function as mlp_chain_bwd_half_kernel (lush_mlp_chain.hip): autograd of NeRF.forward
// (utils/run_lushnerf_helpers.py:394-423) down to d/d(point), d/d(viewdir) through Embedder.forward (:334-344), writing the
// dZ rows the weight-gradient GEMMs read.  Organisation of mlp_wide_fwd_kernel (lush_mlp_wide.hip):
//
//   * a wave owns 64 points = two 32-column MFMA blocks; a layer dZ_{l-1} = (W_l^T dZ_l) * relu'(h_{l-1}) is four QUARTER
//     passes of 64 output rows over the whole K, two accumulator sets in ping-pong: while one accumulates pass p, the
//     other (pass p-1) is masked with the stored ReLU decisions, rounded to fp16 and becomes k-blocks of the next
//     layer's B operand -- as fillers between the MFMAs;
//   * accumulators start from the MFMA's constant-zero C operand (no initialisation), except the four passes behind the
//     feature layer, whose sets are pre-loaded with w_alpha x d_alpha (the alpha head's share of dZ_{NL-1});
//   * the ReLU decisions of a layer (1 KiB per 32-point column block) are fetched by LDS-DMA one layer ahead into a
//     wave-private double buffer, so the conversion reads them with ds_read_u16 and no vmcnt wait stands in an MFMA gap;
//   * the transposed weight stream (NetT::bwd4 copy) is uniform: positions of 8 KiB = 4 k-blocks x 2 row blocks through a
//     ring of WD_S slots, one s_barrier per position.
#include "lush_mlp_wide.h"
#include "lush_host.h"

#include <cstdio>
#include <cstdlib>

namespace lush {

constexpr int WB_WRAP = NetNerf::bwd4_len / 8;      // stream positions per tile
LUSH_CLOCK_DECL(lush_clock_wide_bwd)
#ifdef LUSH_PROF   // developer build: cycle counts (s_memtime) of block 0 / wave 0, read back through lush_debug_prof_wbwd
__device__ unsigned long long lush_prof_wbwd[16];
#define BPROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define BPROF_ADD(slot, t0) cx.prof[slot] += __builtin_amdgcn_s_memtime() - (t0)
#else
#define BPROF_T(var)
#define BPROF_ADD(slot, t0)
#endif
static_assert(NetNerf::bwd4_len % 8 == 0, "the transposed quarter-row stream is whole positions");
constexpr int WB_DPE_LD = 104;                      // 16-bit elements per point in the d(gamma) image (96 used; rows stay 16-byte aligned;
                                                    // 52 dwords: 16 rows at one chunk fall on 16 disjoint bank quads)
// vector-memory operations every tile boundary issues between the last position of one tile and the first of the next: the prefetch
// (2 DMA pairs, 2 loads), the 32 row stores of dZ_0, the 2 head-gradient stores and the 16 row stores of dZv (the d(point) stores of
// a wave that lies beyond P are skipped: not counted)
constexpr int WB_TILE_OPS = 6 + 32 + 2 + 16;
constexpr int WB_SB = 2;                            // stash rows in flight between their LDS read-back and their store
enum { WK_NONE = 0, WK_ACT = 1, WK_ID = 2 };        // pending set: nothing / ReLU mask + fp16 / fp16 only

// Per-pass runtime parameters
struct WbRt {
    unsigned mrd;            // uniform part of the LDS address of the pending set's decision words (column block 0, its first row block;
                             // + c KiB + rbl 128; the per-lane part, 2 lane, comes from the DMA offset register at the point of use)
    const float* initw;      // per-lane LDS pointer: w_alpha of the rows the pending set accumulates next (row block 0, 8 h applied)
    float dalpha[2];         // per-lane: scaled d_alpha of this lane's point in column block c
    const char* msrc[2];     // uniform: the decision words the pass's mask DMA fetches (column block c), minus 1 KiB x wave (the per-lane
                             // offset register of the DMAs is voff = 16 lane + 1024 w)
    unsigned mdst;           // uniform: their LDS address (column block 0; block 1: + 1 KiB)
    char* tile;              // this wave's two 4-KiB LDS transposition tiles
    const char* lut;         // LDS: [256][4] dwords, entry of byte b: dword i = halfword masks of bits 2i, 2i+1 (0xFFFF where the bit is set)
    char* srows;             // uniform: stash rows of this wave's first point
    unsigned srow_off;       // per-lane: byte offset of (row lane>>3, 16-byte chunk lane&7) in a [point][HW] array
};

struct WbConvTmp {
    unsigned w[4];           // decision words of the pending set's four blocks (read in gap 0)
    u32x4 Y[2];              // halfword masks of the 8 values of (block, t): table entry of the word's byte t
    unsigned o[4];           // packed pairs of the current (block, t)
};

// item kinds of a pending set's stream
enum { WI_L = 0, WI_P = 1, WI_ID = 2, WI_IA = 3, WI_IB = 4 };
struct WbItem { int kind, b, t, i; };

// ---------------------------------------------------------------------------------------------
// one quarter pass
// ---------------------------------------------------------------------------------------------
// NRQ = 2: position = 4 k-blocks x 2 row blocks, unit u = (k-block u/2, row block u%2); NRQ = 1: 8 k-blocks x 1 row block.
// Pending set (2 row blocks x 2 column blocks; block b = rbl * 2 + c; (b, t) = its accumulators 8t .. 8t+7 = one k-block of
// the next operand).  The ReLU decisions of (b, t) are one byte; a 4-KiB LDS table turns it into the four halfword-mask
// dwords of its pairs with ONE ds_read_b128, so a pair costs [v_cvt_pk_f16_f32, v_and_b32] (bit tests on the VALU cost
// [shift, and, v_pk_mul_lo_u16] more per pair: with them the fine pass took 3.81 ms, with the table 3.60).  Items:
//   WI_L (b,t)   : table read of (b, t) -- issued one (b, t) ahead of its use, two mask registers sets in ping-pong
//   WI_P (b,t,i) : convert + mask pair i (i = 3 completes the k-block)         WI_ID: convert only (no activation)
//   WI_IA/IB (b,j): pre-load of the set for its next pass: ds_read_b128 of w_alpha into the accumulators / 4 v_mul in place
// poured into the MFMA gaps by the same list scheduler as the forward (a cap of issue slots per gap, deadline D); the four
// decision words are read in gap 0 and the items start in gap 2.
template <int NRQ, int NPOS, int CK, int CKB0, bool INIT, bool ZERO, int D, int PQ, bool STASH, int LD, bool MDMA, int XP = 0>
struct WbPass {
    static constexpr int KBPP = 8 / NRQ;
    static constexpr int NG = NPOS * 16;
    static constexpr int DG = D < NG ? D : NG;
    struct Seq { WbItem it[128]; int n; };
    static constexpr Seq make_seq() {
        Seq q{};
        q.n = 0;
        auto put = [&](int kind, int b, int t, int i) { q.it[q.n].kind = kind; q.it[q.n].b = b; q.it[q.n].t = t; q.it[q.n].i = i; ++q.n; };
        auto init = [&](int b) {
            if (INIT) {
                for (int j = 0; j < 4; ++j) put(WI_IA, b, 0, j);
                for (int j = 0; j < 4; ++j) put(WI_IB, b, 0, j);
            }
        };
        if (CK == WK_ACT) {
            put(WI_L, 0, 0, 0);
            for (int bt = 0; bt < 8; ++bt) {          // (b, t) in order; the table read of the next (b, t) goes first
                if (bt + 1 < 8) put(WI_L, (bt + 1) / 2, (bt + 1) % 2, 0);
                for (int i = 0; i < 4; ++i) put(WI_P, bt / 2, bt % 2, i);
                if (bt % 2 == 1) init(bt / 2);
            }
        } else if (CK == WK_ID) {
            for (int bt = 0; bt < 8; ++bt) {
                for (int i = 0; i < 4; ++i) put(WI_ID, bt / 2, bt % 2, i);
                if (bt % 2 == 1) init(bt / 2);
            }
        } else {
            for (int b = 0; b < 4; ++b) init(b);
        }
        return q;
    }
    static constexpr Seq SQ = make_seq();
    static constexpr int NIT = SQ.n;
    static constexpr int START = CK == WK_ACT ? 2 : 0;      // first gap that takes items
    static_assert(!STASH || NPOS == 4, "the stash pipeline is laid out over the 16 positions of a layer");

    struct Regs {
        bf16x8 a1[4];
        WbConvTmp ct;
        unsigned dma_dst, dma_off;
    };

    // stash pipeline over the layer's 16 positions P = 4 PQ + I, as in the forward: job J = (column block J / 4, k-blocks
    // 4 (J % 4) ..): LDS writes in position J (gaps 12..15, tile J % 2), read-backs in position J + 1, row stores in J + 2 (gaps 4..7)
    static constexpr int ST_JOBS = 8;
    static constexpr bool st_write(int g) { return STASH && g % 16 >= 12 && (4 * PQ + g / 16) < ST_JOBS; }
    static constexpr bool st_read(int g) { return STASH && g % 16 >= 12 && (4 * PQ + g / 16) >= 1 && (4 * PQ + g / 16) <= ST_JOBS; }
    // row i of job J is stored two gaps behind its read-back: gaps 14, 15 of position J + 1 and 0, 1 of position J + 2
    static constexpr int st_store_job(int g) {       // job whose row leaves in gap g, or -1
        const int P = 4 * PQ + g / 16, m = g % 16;
        const int J = m >= 14 ? P - 1 : (m < 2 ? P - 2 : -1);
        return (STASH && J >= 0 && J < ST_JOBS) ? J : -1;
    }
    static constexpr bool st_store(int g) { return st_store_job(g) >= 0; }
    static constexpr bool mdma_at(int g) { return MDMA && g == 6; }

    static constexpr int item_weight(int k) {
        const int kind = SQ.it[k].kind;
        if (kind == WI_L) return 3;                    // byte -> table address (2), ds_read_b128
        if (kind == WI_P) return 3;                    // convert (8 cycles once the gap is full) + and
        if (kind == WI_ID) return 3;
        if (kind == WI_IA) return 1;
        return 4;
    }
    static constexpr int fixed_load(int g) {
        const int m = g % 16;
        int w = 0;
        if (m < 4) w += 1;
        if (m == 4 || m == 5) w += 3;
        if (m == 8) w += 9;
        if (m >= 12) w += 1;
        if (st_write(g)) w += 1;
        if (st_read(g)) w += 1;
        if (st_store(g)) w += 2;
        if (mdma_at(g)) w += 8;
        if (CK == WK_ACT && g == 0) w += 4;
        return w;
    }
    struct Sched {
        int first[NG + 1];
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
                        if (g < (k == 0 && CK == WK_ACT ? START - 1 : START)) break;     // (the first table read goes one gap ahead)
                        const int w = item_weight(k);
                        if (load + w > cap && !(g == NG - 1)) break;
                        load += w;
                        ++k;
                    }
                }
            }
            S.first[NG] = k;
            S.cap = cap;
            int last = 0;
            for (int g = 0; g < NG; ++g)
                if (S.first[g + 1] > S.first[g]) last = g;
            if (k == NIT && (last < DG || NIT == 0)) return S;
        }
        return S;
    }
    static constexpr Sched SC = make_sched();

    template <int I, int M>
    static __device__ __forceinline__ void mfma(f32x16 (&act)[2][2], const Regs& r, const WdCarry& cr, const u32x4 (&xin)[2][16]) {
        constexpr int u = M / 2, c = M % 2;
        constexpr int kbl = NRQ == 2 ? u / 2 : u, rbl = NRQ == 2 ? u % 2 : 0;
        const bf16x8 a = u < 4 ? cr.a0[u] : r.a1[u - 4];
        const bf16x8 b = __builtin_bit_cast(bf16x8, xin[c][I * KBPP + kbl]);
        if constexpr (ZERO && I == 0 && kbl == 0) {
            f32x16 z;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = 0.f;
            act[c][rbl] = mfma_f16(a, b, z);
        } else {
            act[c][rbl] = mfma_f16(a, b, act[c][rbl]);
        }
    }

    // (2 lane + 128 w = voff / 8, formed here and dead after the four reads: as a per-lane pointer held through the tile it was
    // the value the register allocator spilled, and a scratch reload waits vmcnt(0))
    static __device__ __forceinline__ void load_words(Regs& r, const WbRt& rt, unsigned voff) {
        unsigned a;
        asm volatile("v_lshrrev_b32 %0, 3, %1\n\tv_add_u32 %0, %0, %2" : "=&v"(a) : "v"(voff), "s"(rt.mrd));
        typedef const __attribute__((address_space(3))) unsigned short* lds_u16;
#pragma unroll
        for (int b = 0; b < 4; ++b) r.ct.w[b] = *reinterpret_cast<lds_u16>(a + (b % 2) * 1024 + (b / 2) * 128);
    }

    template <int I, int M>
    static __device__ __forceinline__ void fixed(WdCtx& cx, Regs& r, WdCarry& cr, const char* rd, const char* rd_next, const WbRt& rt) {
        if constexpr (M < 4) {
            r.a1[M] = *reinterpret_cast<const bf16x8*>(rd + (4 + M) * 1024 + cx.voff);
            if constexpr (CK == WK_ACT && I == 0 && M == 0) load_words(r, rt, cx.voff);
        } else if constexpr (M == 4) {
            r.dma_dst = cx.dma_base + cx.slot_off;              // refill the slot this position frees ...
            r.dma_off = cx.fetch_off;                           // ... with stream position +S
        } else if constexpr (M == 5) {
            cx.fetch_off = cx.fetch_off + WD_SLOT == (unsigned)WB_WRAP * WD_SLOT ? 0u : cx.fetch_off + WD_SLOT;
            cx.slot_off = cx.slot_off + WD_SLOT == (unsigned)WD_S * WD_SLOT ? 0u : cx.slot_off + WD_SLOT;
        } else if constexpr (M == 6) {
            if constexpr (MDMA && I == 0) wd_dma_pair(wd_uniform(rt.msrc[0]), wd_uniform(rt.msrc[1]), cx.voff, __builtin_amdgcn_readfirstlane(rt.mdst), __builtin_amdgcn_readfirstlane(rt.mdst + 1024u));
        } else if constexpr (M == 8) {
            wd_dma_pair(cx.gbase + r.dma_off, cx.gbase + r.dma_off + 4096u, cx.voff, r.dma_dst, r.dma_dst + 4096u);
        } else if constexpr (M >= 12) {      // (four gaps ahead of their first MFMA, not eight: 8 fewer live registers at the peak)
            cr.a0[M - 12] = *reinterpret_cast<const bf16x8*>(rd_next + (M - 12) * 1024 + cx.voff);
        }
    }

    // item K of the pending set's stream.  Conversion units are volatile asm: pure VALU code is otherwise free to leave its
    // filler slot (the compiler gathers it at the head of the basic block, where nothing hides it).
    template <int K>
    static __device__ __forceinline__ void item(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], Regs& r, const WbRt& rt) {
        constexpr WbItem it = SQ.it[K];
        constexpr int b = it.b, t = it.t, c = b % 2, rbl = b / 2;
        if constexpr (it.kind == WI_L) {
            unsigned a;
#ifdef LUSH_ABL_LUTFREE      // timing ablation only (wrong masks): every lane of a 16-lane group reads its own bank quad -- what the table reads' bank
            // conflicts cost.  Entry 17 x (lane & 15) = 0x00, 0x11 .. 0xFF: bank quad lane & 15, and HALF of the decisions set on average,
            // as in the real masks (entries 0 .. 15 would zero three quarters of dZ: a lighter operand stream, a higher clock)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_and_b32 %0, 15, %0\n\tv_mul_u32_u24 %0, 17, %0" : "=v"(a) : "v"(r.ct.w[b]));
#else
            asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(a) : "v"(r.ct.w[b]), "n"(8 * t));
#endif
            r.ct.Y[(2 * b + t) % 2] = *reinterpret_cast<const u32x4*>(rt.lut + a * 16);
        } else if constexpr (it.kind == WI_P || it.kind == WI_ID) {
            constexpr int i = it.i;
            if constexpr (it.kind == WI_P)
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_and_b32 %0, %0, %3"
                             : "=&v"(r.ct.o[i]) : "v"(pend[c][rbl][8 * t + 2 * i]), "v"(pend[c][rbl][8 * t + 2 * i + 1]), "v"(r.ct.Y[(2 * b + t) % 2][i]));
            else
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.ct.o[i]) : "v"(pend[c][rbl][8 * t + 2 * i]), "v"(pend[c][rbl][8 * t + 2 * i + 1]));
            if constexpr (i == 3) {      // the k-block leaves as one 128-bit value
                const u32x4 v = {r.ct.o[0], r.ct.o[1], r.ct.o[2], r.ct.o[3]};
                xout[c][CKB0 + 2 * rbl + t] = v;
            }
        } else {
            constexpr int j = it.i;       // accumulators 4j .. 4j+3 = rows (0, 4, 16, 20)[j] + 8 h + e of the block
            if constexpr (it.kind == WI_IA) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rt.initw + rbl * 32 + (j == 0 ? 0 : j == 1 ? 4 : j == 2 ? 16 : 20));
#pragma unroll
                for (int e = 0; e < 4; ++e) pend[c][rbl][4 * j + e] = v[e];
            } else {
                // Four plain v_mul_f32 as one volatile statement.  Written as C++ the compiler pairs them into v_pk_mul_f32, and with
                // the last v_pk_mul_f32 of a block directly in front of a run of MFMAs its result was lost in lanes 48..63
                // (measured: d_alpha's share missing in accumulators 14 / 15 of those lanes, in whichever waves ran that way;
                // DESIGN.md section 4) -- the plain form is right and stays where the schedule puts it.
                asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
                             : "+v"(pend[c][rbl][4 * j]), "+v"(pend[c][rbl][4 * j + 1]), "+v"(pend[c][rbl][4 * j + 2]), "+v"(pend[c][rbl][4 * j + 3])
                             : "v"(rt.dalpha[c]));
            }
        }
    }

    template <int G>
    static __device__ __forceinline__ void pending(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], Regs& r, const WbRt& rt) {
#ifdef LUSH_ABL_NOCONV      // timing ablation only (wrong results)
        return;
#endif
        wd_unroll<SC.first[G], SC.first[G + 1]>([&](auto kc) __attribute__((always_inline)) { item<decltype(kc)::value>(pend, xout, r, rt); });
    }

    // one 16-byte piece per lane: scalar row base + ONE per-lane offset register (as C++ the four rows of a job, 4 KiB apart, each
    // held a 64-bit per-lane address: 8 registers this kernel does not have -- the loop then spilled, and every scratch reload
    // waits vmcnt(0), i.e. for every stash store in flight)
    template <int JOB, int I4>
    static __device__ __forceinline__ void stash_store(const WdCarry& cr, const WbRt& rt) {
        constexpr int c = JOB / 4, j = JOB % 4;
#ifndef LUSH_ABL_NOSTORE
        // (s_nop 1: a store of more than 8 bytes reads its data registers for two more cycles -- the hazard the compiler covers for its
        // own stores; without it the conversion's VALU code, which re-uses the row's registers at once, reached memory in some lanes)
#ifdef LUSH_PLAIN_STASH      // developer A/B: cached stores
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(rt.srow_off), "v"(cr.sb[I4 % WB_SB]), "s"(rt.srows + ((c * 32 + 8 * I4) * LD + j * 64) * 2) : "memory");
#else
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(rt.srow_off), "v"(cr.sb[I4 % WB_SB]), "s"(rt.srows + ((c * 32 + 8 * I4) * LD + j * 64) * 2) : "memory");
#endif
#else
        asm volatile("" ::"v"(cr.sb[I4 % WB_SB]));
#endif
    }
    template <int G>
    static __device__ __forceinline__ void stash(const u32x4 (&xin)[2][16], WdCarry& cr, const WbRt& rt, int lane) {
#ifdef LUSH_ABL_NOSTASH
        return;
#endif
        constexpr int P = 4 * PQ + G / 16, m = G % 16;
        if constexpr (st_store(G)) stash_store<st_store_job(G), (m >= 14 ? m - 14 : m + 2)>(cr, rt);
        if constexpr (st_write(G)) {
            constexpr int job = P, c = job / 4, j = job % 4, o = m - 12;
            const int n = lane & 31, hh = lane >> 5;
            *reinterpret_cast<u32x4*>(rt.tile + (job % 2) * 4096 + n * 128 + (((2 * o + hh) ^ (n & 7)) << 4)) = xin[c][4 * j + o];
        }
        if constexpr (st_read(G)) {
            constexpr int job = P - 1, i = m - 12;
            const int row = 8 * i + (lane >> 3);
            cr.sb[i % WB_SB] = *reinterpret_cast<const u32x4*>(rt.tile + (job % 2) * 4096 + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        }
    }

    // vector-memory operations this wave issues in main gaps [g0, g1) besides the refill pairs (vmcnt bookkeeping; only what
    // is certain: an undercount is safe)
    static constexpr int others_in(int g0, int g1) {
        int n = 0;
        for (int g = g0 < 0 ? 0 : g0; g < g1 && g < NG; ++g) {
            if (st_store(g)) ++n;
            if (mdma_at(g)) n += 2;
        }
        return n;
    }

    template <int I>
    static __device__ __forceinline__ void position(WdCtx& cx, f32x16 (&act)[2][2], f32x16 (&pend)[2][2], const u32x4 (&xin)[2][16],
                                                    u32x4 (&xout)[2][16], Regs& r, WdCarry& cr, const WbRt& rt) {
        const char* rd = cx.ring + cx.slot_off;           // (cx.ring is the ring minus 1 KiB x wave: the per-lane part is the DMA offset)
        __builtin_amdgcn_sched_barrier(0);
        wd_unroll<0, 8>([&](auto mc) __attribute__((always_inline)) {
            constexpr int M = decltype(mc)::value;
            mfma<I, M>(act, r, cr, xin);
            fixed<I, M>(cx, r, cr, rd, nullptr, rt);
            stash<16 * I + M>(xin, cr, rt, cx.lane);
            pending<16 * I + M>(pend, xout, r, rt);
            __builtin_amdgcn_sched_barrier(0);
        });
        // mid-step: my pieces of position +1 have landed (the DMAs of +2 .. +S-1 and what was issued since are younger);
        // after the barrier everyone's have, and nobody reads this position's slot any more
        // (XP: the pass's first XP positions are among the first S - 1 of a tile: the piece awaited was requested in the previous
        // tile, in front of everything the tile boundary issues)
        constexpr int younger = 2 * (WD_S - 2) + others_in(16 * (I + 1 - WD_S) + 8, 16 * I + 8) + (I < XP ? WB_TILE_OPS : 0);
        BPROF_T(t_w0);
#ifndef LUSH_ABL_NOVMWAIT
        wd_wait_vm<(younger < 63 ? younger : 63)>();
#endif
        BPROF_T(t_w1);
        lds_barrier();
#ifdef LUSH_PROF
        cx.prof[5] += t_w1 - t_w0;
        cx.prof[6] += __builtin_amdgcn_s_memtime() - t_w1;
        cx.prof[7] += 1;
#endif
        const char* rd_next = cx.ring + cx.slot_off;        // (slot_off already names the next slot: gap 5)
        __builtin_amdgcn_sched_barrier(0);
        wd_unroll<8, 16>([&](auto mc) __attribute__((always_inline)) {
            constexpr int M = decltype(mc)::value;
            mfma<I, M>(act, r, cr, xin);
            fixed<I, M>(cx, r, cr, rd, rd_next, rt);
            stash<16 * I + M>(xin, cr, rt, cx.lane);
            pending<16 * I + M>(pend, xout, r, rt);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    static __device__ __forceinline__ void run(WdCtx& cx, f32x16 (&act)[2][2], f32x16 (&pend)[2][2], const u32x4 (&xin)[2][16],
                                               u32x4 (&xout)[2][16], WdCarry& cr, const WbRt& rt) {
        Regs r;
        wd_unroll<0, NPOS>([&](auto ic) __attribute__((always_inline)) { position<decltype(ic)::value>(cx, act, pend, xin, xout, r, cr, rt); });
    }

    // the pending set's whole item stream with no MFMAs to hide behind (tile prologue)
    static __device__ __forceinline__ void convert_now(f32x16 (&pend)[2][2], u32x4 (&xout)[2][16], const WbRt& rt, unsigned voff) {
        Regs r;
        if constexpr (CK == WK_ACT) load_words(r, rt, voff);
        wd_unroll<0, NIT>([&](auto kc) __attribute__((always_inline)) { item<decltype(kc)::value>(pend, xout, r, rt); });
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <class N>
__global__ __launch_bounds__(WD_NT) void mlp_wide_bwd_kernel(const MlpBwdArgs A) {
    static_assert(N::HW == 256 && N::NL == 8 && N::HV == 128 && N::SKIP == 5, "the wide kernel is built for the 8x256 net");
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL;
    constexpr int LDV = HV + DZV_EXT;                          // dZv rows carry the head gradients in 8 more columns
    constexpr int CB_BYTES = N::n_mask_layers * 1024;          // decision words of one column block: 1 KiB per mask layer
    LUSH_CLOCK_STAMP(lush_clock_wide_bwd, 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wtab = reinterpret_cast<float*>(smem);              // w_rgb [3][HV] | w_alpha [HW]   (first: DS immediates reach it)
    constexpr int WTAB_BYTES = (3 * HV + HW) * 4;
    char* ring = smem + WTAB_BYTES;                            // [WD_S][8 KiB]
    char* stage = ring + WD_S * WD_SLOT;                       // [4 waves][2][4 KiB] stash transposition tiles
    char* maskbuf = stage + 8 * 4096;                          // [4 waves][2 parities][2 column blocks][1 KiB]
    __bf16* dpe = reinterpret_cast<__bf16*>(maskbuf + 4 * 4096);   // [256 points][WB_DPE_LD] d(gamma), fp16
    unsigned* lut = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dpe) + WD_MT * WB_DPE_LD * 2);   // [256][4]: halfword masks of a decision byte

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const char* wbase = reinterpret_cast<const char*>(A.wpk);
    {
        const float* f32 = reinterpret_cast<const float*>(wbase + (long long)N::total_entries * 1024);
        for (int i = tid; i < 3 * HV + HW; i += WD_NT) wtab[i] = f32[N::f32_w_rgb + i];
    }
    for (int i = threadIdx.x; i < 1024; i += WD_NT) {
        const int b = i >> 2, p2 = 2 * (i & 3);
        lut[i] = (((b >> p2) & 1) ? 0x0000FFFFu : 0u) | (((b >> (p2 + 1)) & 1) ? 0xFFFF0000u : 0u);
    }
    const float* w_rgb = wtab;
    const float* w_alpha = wtab + 3 * HV;
    const float gscale = A.scale[0], ginv = A.scale[1];
    // a live-point launch (MlpBwdArgs::live_idx) reads its point count on the device
    int P = A.P, n_tiles = A.n_tiles;
    if (A.live_cnt != nullptr) {
        P = __builtin_amdgcn_readfirstlane(*A.live_cnt);
        n_tiles = (P + WD_MT - 1) / WD_MT;
    }

    WdCtx cx;
    cx.ring = ring - w * 1024;      // fragment reads add voff = 16 lane + 1024 w: one per-lane register serves the DMAs and the reads
    cx.ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    cx.gbase = wbase + (long long)N::bwd4_base * 1024;
    cx.slot_off = 0;
    cx.fetch_off = (unsigned)(WD_S % WB_WRAP) * WD_SLOT;
    cx.dma_base = __builtin_amdgcn_readfirstlane(cx.ring_lds + (unsigned)w * 1024u);
    cx.w = w;
    cx.lane = lane;
    cx.voff = (unsigned)lane * 16u + (unsigned)w * 1024u;
#pragma unroll
    for (int j = 0; j < WD_S; ++j)
        wd_dma_pair(cx.gbase + (unsigned)j * WD_SLOT, cx.gbase + (unsigned)j * WD_SLOT + 4096u, cx.voff, cx.dma_base + (unsigned)j * WD_SLOT, cx.dma_base + (unsigned)j * WD_SLOT + 4096u);

    const int row0 = w * 64;
    char* const mbuf_w = maskbuf + w * 4096;                                   // this wave's [2 parities][2][1 KiB]
    const unsigned mbuf_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)mbuf_w);
    WbRt rt;
    rt.tile = stage + w * 8192;
    rt.srows = nullptr;
    rt.srow_off = (unsigned)(((lane >> 3) * HW + (lane & 7) * 8) * 2);
    rt.lut = reinterpret_cast<const char*>(lut);
    rt.initw = w_alpha + 8 * h;
    const unsigned mrd0 = mbuf_lds - (unsigned)w * 128u;      // + 2 lane + 128 w (= voff / 8) = this lane's word in the wave's buffer
    rt.mrd = mrd0;
    // d(gamma) columns 32 b + 16 t + 8 h + (0..7) of this lane's point in column block c
    __bf16* const grow0 = dpe + (row0 + n) * WB_DPE_LD + 8 * h;
    auto dpe_put = [&](const f32x16& a, int c, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned o[1];
                split_pair<1, DT_F16>(a[8 * t + 2 * i], a[8 * t + 2 * i + 1], o);
                pk[i] = o[0];
            }
            *reinterpret_cast<u32x4*>(grow0 + c * 32 * WB_DPE_LD + 32 * b + 16 * t) = pk;
        }
    };
    auto dpe_add = [&](f32x16& a, int c, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(grow0 + c * 32 * WB_DPE_LD + 32 * b + 16 * t);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[8 * t + i] += elem_to_f32<DT_F16>(v[i]);
        }
    };
    // 4 k-blocks of one column block -> 32 rows x 128 bytes of a [point][LD_] array, through the wave's LDS tile; the stores go out
    // through a scalar row base + one per-lane offset (as C++ every row held a 64-bit per-lane address, spilled and reloaded
    // with vmcnt(0) at the head of every tile)
    auto stash_now = [&](const u32x4 (&x)[2][16], int c, int j, const char* rows, int LD_) __attribute__((always_inline)) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
            *reinterpret_cast<u32x4*>(rt.tile + n * 128 + (((2 * kq + h) ^ (n & 7)) << 4)) = x[c][4 * j + kq];
        const unsigned voff_row = (unsigned)(((lane >> 3) * LD_ + (lane & 7) * 8) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3);
            const u32x4 v = *reinterpret_cast<const u32x4*>(rt.tile + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff_row), "v"(v), "s"(rows + ((c * 32 + 8 * i) * LD_ + j * 64) * 2) : "memory");
        }
    };

    // What a tile needs before its first MFMA -- d_raw of the lane's two points and the decision words of the views hidden and of
    // h_{NL-1} -- is fetched while the PREVIOUS tile runs its epilogue (first tile: here), as volatile asm so that this kernel,
    // not the compiler, places the wait: the tile start then waits for these loads with the dZ_0 rows of the previous tile
    // still in flight behind them, instead of draining the queue (measured: prologue + epilogue were 20 % of the launch).
    f32x4 drn[2];
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        const long long wp = (long long)t * WD_MT + row0;
        const char* mb0 = wd_uniform(reinterpret_cast<const char*>(A.mask) + (wp / 32) * CB_BYTES - w * 1024);
        wd_dma_pair(mb0 + NL * 1024, mb0 + CB_BYTES + NL * 1024, cx.voff, mbuf_lds + (NL % 2) * 2048u, mbuf_lds + (NL % 2) * 2048u + 1024u);
        wd_dma_pair(mb0 + (NL - 1) * 1024, mb0 + CB_BYTES + (NL - 1) * 1024, cx.voff, mbuf_lds + ((NL - 1) % 2) * 2048u, mbuf_lds + ((NL - 1) % 2) * 2048u + 1024u);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            long long gpt = wp + c * 32 + n;
            if (gpt >= P) gpt = P - 1;                        // (zeroed when consumed)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(drn[c]) : "v"((unsigned)(gpt * 16)), "s"(A.draw) : "memory");
        }
    };
    // (first tile: waited for right here -- nothing runs beside it anyway -- so that the tile loop's head has ONE wait for every
    // tile.  With a vmcnt(0) for the first tile and a vmcnt(32) for the others selected by `tile == blockIdx.x` at the loop
    // head, no path-insensitive reading of the code can tell that the prefetch is always waited for; lush_nerf_amd/isa_check.py
    // rule R5 follows every path and now finds the wait on each of them.)
    if ((int)blockIdx.x < n_tiles) prefetch(blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(drn[0]), "+v"(drn[1])::"memory");
#ifdef LUSH_PROF
    for (int i = 0; i < 16; ++i) cx.prof[i] = 0;
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long pt0 = (long long)tile * WD_MT;
        const long long wpt = pt0 + row0;
        BPROF_T(t_tile);
        {   // opaque per tile (keeps the static stream addresses from being hoisted out of the tile loop)
            unsigned long long gb = (unsigned long long)cx.gbase;
            asm volatile("" : "+s"(gb));
            cx.gbase = (const char*)gb;
        }
        // decision words of (this wave's column block c, mask layer ml): mbase[c] + ml KiB (+ 1 KiB x wave: DMA sources only)
        const char* mbase[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) mbase[c] = wd_uniform(reinterpret_cast<const char*>(A.mask) + (wpt / 32 + c) * CB_BYTES - w * 1024);
        // the prefetch has landed.  First tile: waited for in front of the loop; later tiles: the 32 row stores of dZ_0 were issued
        // behind it (and the two d(point) stores, unless the whole wave lies beyond P: not counted -- an undercount is safe)
        asm volatile("s_waitcnt vmcnt(32)" : "+v"(drn[0]), "+v"(drn[1])::"memory");
        float4 dr[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool in = wpt + c * 32 + n < P;
            dr[c].x = in ? drn[c][0] * gscale : 0.f; dr[c].y = in ? drn[c][1] * gscale : 0.f;
            dr[c].z = in ? drn[c][2] * gscale : 0.f; dr[c].w = in ? drn[c][3] * gscale : 0.f;
            rt.dalpha[c] = dr[c].w;
        }
        BPROF_ADD(1, t_tile);      // wait for the prefetch
        BPROF_T(t_bar);
        lds_barrier();         // the first position's pieces of every wave have landed (and wtab, first tile)
        BPROF_ADD(2, t_bar);
        BPROF_T(t_pro);

        f32x16 accA[2][2], accB[2][2];
        u32x4 B0[2][16], B1[2][16];
        WdCarry a0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a0.a0[i] = *reinterpret_cast<const bf16x8*>(cx.ring + cx.slot_off + i * 1024 + cx.voff);

#ifndef LUSH_ABL_NOPRO      // timing ablation only (wrong results)
        // ---- dZv = (Wrgb^T d_rgb) * relu'(hv): K = 3, a rank-3 update on the VALU; rows 0..63 -> set A, 64..127 -> set B ----
        {
            auto rank3 = [&](f32x16 (&acc)[2][2], int r0) __attribute__((always_inline)) {
#pragma unroll
                for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = r0 + rbl * 32 + 8 * h + (j == 0 ? 0 : j == 1 ? 4 : j == 2 ? 16 : 20);
                        const f32x4 wr = *reinterpret_cast<const f32x4*>(w_rgb + f);
                        const f32x4 wg = *reinterpret_cast<const f32x4*>(w_rgb + HV + f);
                        const f32x4 wb = *reinterpret_cast<const f32x4*>(w_rgb + 2 * HV + f);
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[c][rbl][4 * j + e] = wr[e] * dr[c].x + wg[e] * dr[c].y + wb[e] * dr[c].z;
                    }
            };
            rank3(accA, 0);
            rank3(accB, 64);
            const unsigned m8 = mrd0 + (NL % 2) * 2048;
            rt.mrd = m8;
            WbPass<2, 1, WK_ACT, 0, false, true, 16, 0, false, HW, false>::convert_now(accA, B1, rt, cx.voff);
            rt.mrd = m8 + 2 * 128;
            WbPass<2, 1, WK_ACT, 4, false, true, 16, 0, false, HW, false>::convert_now(accB, B1, rt, cx.voff);
        }
        // the four head gradients of the lane's points as a hi and a lo 16-bit plane in the 8 extra columns of their dZv rows
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 d = dr[c];
                unsigned a[1], b[1], e[1], f[1];
                split_pair<1, DT_F16>(d.x, d.y, a);
                split_pair<1, DT_F16>(d.z, d.w, b);
                const float hx = elem_to_f32<DT_F16>(__builtin_bit_cast(__bf16, (unsigned short)(a[0] & 0xFFFFu)));
                const float hy = elem_to_f32<DT_F16>(__builtin_bit_cast(__bf16, (unsigned short)(a[0] >> 16)));
                const float hz = elem_to_f32<DT_F16>(__builtin_bit_cast(__bf16, (unsigned short)(b[0] & 0xFFFFu)));
                const float hw = elem_to_f32<DT_F16>(__builtin_bit_cast(__bf16, (unsigned short)(b[0] >> 16)));
                split_pair<1, DT_F16>(d.x - hx, d.y - hy, e);
                split_pair<1, DT_F16>(d.z - hz, d.w - hw, f);
                u32x4 v;
                v[0] = a[0]; v[1] = b[0]; v[2] = e[0]; v[3] = f[0];
                *reinterpret_cast<u32x4*>(A.dzv + (wpt + c * 32 + n) * LDV + HV) = v;
            }
        }
        {   // dZv rows
            const char* rows = wd_uniform(reinterpret_cast<const char*>(A.dzv + wpt * LDV));
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < N::KKV / 4; ++j) stash_now(B1, c, j, rows, LDV);
        }
#endif
        // ---- d_feature = Wva^T dZv (four quarters of 2 positions); d gamma(d) = Wvb^T dZv (one position, one row block) ----
        BPROF_ADD(3, t_pro);       // rank-3 update, conversion, dZv rows
        BPROF_T(t_body);
        if (tile == (int)blockIdx.x) wd_wait_vm<0>();      // (first tile: no previous tile issued what the counts below assume)
        static_assert(WD_S - 1 == 5, "the first S - 1 positions of a tile: VA0, VA1 and the first of VA2");
        WbPass<2, 2, WK_NONE, 0, false, true, 32, 0, false, HW, false, 2>::run(cx, accA, accB, B1, B0, a0, rt);
        WbPass<2, 2, WK_ID, 0, false, true, 28, 0, false, HW, false, 2>::run(cx, accB, accA, B1, B0, a0, rt);
        WbPass<2, 2, WK_ID, 4, false, true, 28, 0, false, HW, false, 1>::run(cx, accA, accB, B1, B0, a0, rt);
        WbPass<2, 2, WK_ID, 8, false, true, 28, 0, false, HW, false>::run(cx, accB, accA, B1, B0, a0, rt);
        WbPass<1, 1, WK_ID, 12, false, true, 16, 0, false, HW, false>::run(cx, accA, accB, B1, B0, a0, rt);
#pragma unroll
        for (int c = 0; c < 2; ++c) dpe_put(accA[c][0], c, 2);
        // ---- dZ_{NL-1} = (Wfeat^T d_feature + Walpha^T d_alpha) * relu'(h_{NL-1}): sets pre-loaded with the alpha share ----
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(w_alpha + 8 * h + rbl * 32 + (j == 0 ? 0 : j == 1 ? 4 : j == 2 ? 16 : 20));
#pragma unroll
                    for (int e = 0; e < 4; ++e) accA[c][rbl][4 * j + e] = v[e] * rt.dalpha[c];
                }
        {
            const unsigned m7 = mrd0 + ((NL - 1) % 2) * 2048;
            rt.initw = w_alpha + 8 * h + 64;
            WbPass<2, 4, WK_NONE, 0, true, false, 60, 0, false, HW, false>::run(cx, accA, accB, B0, B1, a0, rt);
            rt.initw = w_alpha + 8 * h + 128; rt.mrd = m7;
            rt.msrc[0] = mbase[0] + (NL - 2) * 1024; rt.msrc[1] = mbase[1] + (NL - 2) * 1024; rt.mdst = mbuf_lds + ((NL - 2) % 2) * 2048u;
            WbPass<2, 4, WK_ACT, 0, true, false, 60, 1, false, HW, true>::run(cx, accB, accA, B0, B1, a0, rt);
            rt.initw = w_alpha + 8 * h + 192; rt.mrd = m7 + 2 * 128;
            WbPass<2, 4, WK_ACT, 4, true, false, 60, 2, false, HW, false>::run(cx, accA, accB, B0, B1, a0, rt);
            rt.mrd = m7 + 4 * 128;
            WbPass<2, 4, WK_ACT, 8, false, false, 60, 3, false, HW, false>::run(cx, accB, accA, B0, B1, a0, rt);
        }
        // ---- trunk: dZ_{l-1} = (W_l^T dZ_l) * relu'(h_{l-1}), l = NL-1 .. 1; dZ_l rows leave while they are the B operand ----
        auto layer = [&](int l, u32x4 (&xin)[2][16], u32x4 (&xout)[2][16]) __attribute__((always_inline)) {
            const unsigned ml = mrd0 + (unsigned)(l % 2) * 2048u;            // decisions of h_l (the pending last quarter of dZ_l)
            const unsigned mo = mrd0 + (unsigned)((l - 1) % 2) * 2048u;      // decisions of h_{l-1}
            rt.srows = const_cast<char*>(wd_uniform(reinterpret_cast<const char*>(A.dz0 + (long long)l * A.dz_stride + wpt * HW)));
            rt.mrd = ml + 6 * 128;
            WbPass<2, 4, WK_ACT, 12, false, true, 48, 0, true, HW, false>::run(cx, accA, accB, xin, xin, a0, rt);
            rt.mrd = mo;
            {   // the next layer's decisions (h_{l-2}) into the parity h_l just left; l = 1 re-fetches h_0 (nobody reads it)
                const int mn = l >= 2 ? l - 2 : 0;
                rt.msrc[0] = mbase[0] + mn * 1024; rt.msrc[1] = mbase[1] + mn * 1024; rt.mdst = mbuf_lds + (unsigned)(l % 2) * 2048u;
            }
            WbPass<2, 4, WK_ACT, 0, false, true, 60, 1, true, HW, true>::run(cx, accB, accA, xin, xout, a0, rt);
            rt.mrd = mo + 2 * 128;
            WbPass<2, 4, WK_ACT, 4, false, true, 60, 2, true, HW, false>::run(cx, accA, accB, xin, xout, a0, rt);
            rt.mrd = mo + 4 * 128;
            WbPass<2, 4, WK_ACT, 8, false, true, 60, 3, true, HW, false>::run(cx, accB, accA, xin, xout, a0, rt);
        };
#pragma unroll 1
        for (int l2 = 0; l2 < NL / 2; ++l2) {
            const int l = NL - 1 - 2 * l2;                 // 7, 5, 3, 1
            layer(l, B1, B0);
            if (l == N::SKIP) {   // gamma(x) rows of the skip layer's input (set A is free, B keeps the pending quarter), parked in
                                  // the image until layer 0 adds its share
                WbPass<2, 4, WK_NONE, 0, false, true, 64, 0, false, HW, false>::run(cx, accA, accB, B1, B1, a0, rt);
#pragma unroll
                for (int c = 0; c < 2; ++c) { dpe_put(accA[c][0], c, 0); dpe_put(accA[c][1], c, 1); }
            }
            if (l > 1) layer(l - 1, B0, B1);               // 6, 4, 2
        }
        // ---- layer 0: d gamma(x) += W_0^T dZ_0; set B (last quarter of dZ_0) -> k-blocks 12..15 ----
        // the point this THREAD differentiates the encoding at (point row0 + lane) is loaded in front of the pass: B1 is dead, and
        // the loads have four positions to land
        float px[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
        if (wpt + lane < P) point_of(A.rays, A.z, A.S, A.live_idx ? (long long)A.live_idx[wpt + lane] : wpt + lane, px, pd);
        rt.mrd = mrd0 + 6 * 128;                           // h_0: parity 0
        WbPass<2, 4, WK_ACT, 12, false, true, 48, 0, false, HW, false>::run(cx, accA, accB, B0, B0, a0, rt);
        BPROF_ADD(4, t_body);      // VA .. layer 0
        BPROF_T(t_epi);
#ifndef LUSH_ABL_NOEPI      // timing ablation only (wrong results)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            dpe_add(accA[c][0], c, 0); dpe_add(accA[c][1], c, 1);
            dpe_put(accA[c][0], c, 0); dpe_put(accA[c][1], c, 1);
        }
        // (the point is consumed here, in front of the prefetch: the compiler's wait for its loads would otherwise cover the
        // prefetch too, which it cannot count)
        float rvh[3], rvl[3], rdh[3], rdl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rev_split(px[i], &rvh[i], &rvl[i]);
            rev_split(pd[i], &rdh[i], &rdl[i]);
            asm volatile("" : "+v"(rvh[i]), "+v"(rvl[i]), "+v"(rdh[i]), "+v"(rdl[i]));
        }
        // (parities 0 / 1 of the decision buffer are free now.  Unconditional -- the last tile fetches its own inputs once more --
        // so that d_raw's registers are re-defined on every path and dead through the tile body, not carried across it)
        prefetch(tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile);
        {   // dZ_0 rows: behind the prefetch (the tile start counts them as younger), in front of the encoding (B0's 128 registers are
            // free there: with B0 alive through it the allocator spilled the per-lane DMA offset for the whole kernel)
            const char* rows = wd_uniform(reinterpret_cast<const char*>(A.dz0 + wpt * HW));
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) stash_now(B0, c, j, rows, HW);
        }
        BPROF_ADD(9, t_epi);       // image, prefetch, dZ_0 rows
        BPROF_T(t_enc);
        // ---- through the encoding: d/dx_i = g[i] + sum_k 2^k (cos(2^k x_i) g_sin - sin(2^k x_i) g_cos), one point per thread.
        // The rows of this wave's 64 points were written by this wave alone: LDS operations of a wave execute in order.
        {
            const __bf16* g = dpe + (row0 + lane) * WB_DPE_LD;
            float gv[PE_X + PE_D];
#pragma unroll
            for (int cc = 0; cc < (PE_X + PE_D) / 8; ++cc) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(g + 8 * cc);
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[8 * cc + e] = elem_to_f32<DT_F16>(v[e]);
            }
            float gx[3], gd[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float hx = rvh[i], lx = rvl[i], hd = rdh[i], ld = rdl[i];
                float sx = gv[i], sd = gv[PE_X + i];
#pragma unroll
                for (int k = 0; k < L_X; ++k) {
                    float sn, cs;
                    sincos_rev(hx, lx, k, &sn, &cs);
                    sx += (float)(1 << k) * (cs * gv[3 + 6 * k + i] - sn * gv[3 + 6 * k + 3 + i]);
                }
#pragma unroll
                for (int k = 0; k < L_D; ++k) {
                    float sn, cs;
                    sincos_rev(hd, ld, k, &sn, &cs);
                    sd += (float)(1 << k) * (cs * gv[PE_X + 3 + 6 * k + i] - sn * gv[PE_X + 3 + 6 * k + 3 + i]);
                }
                gx[i] = sx * ginv;
                gd[i] = sd * ginv;
            }
            if (wpt + lane < P) {
                float4* o = reinterpret_cast<float4*>(A.dpts + (wpt + lane) * 8);
                o[0] = make_float4(gx[0], gx[1], gx[2], 0.f);
                o[1] = make_float4(gd[0], gd[1], gd[2], 0.f);
            }
        }
#endif
        BPROF_ADD(10, t_enc);
#ifdef LUSH_PROF
        cx.prof[8] += 1;
#endif
    }
#ifdef LUSH_PROF
    if (blockIdx.x == 0 && tid == 0) {
        cx.prof[0] = __builtin_amdgcn_s_memtime() - t_kernel;
        for (int i = 0; i < 16; ++i) lush_prof_wbwd[i] = cx.prof[i];
    }
#endif
    wd_wait_vm<0>();          // the look-ahead DMAs of the non-existent next tile must land before the LDS is released
    LUSH_CLOCK_STAMP(lush_clock_wide_bwd, 1);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
LUSH_CLOCK_EXPORT(lush_debug_clock_chain, lush_clock_wide_bwd)
#ifdef LUSH_PROF
extern "C" int lush_debug_prof_wbwd(unsigned long long* out) {
    LUSH_HIP(hipDeviceSynchronize());
    LUSH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(lush_prof_wbwd), sizeof(unsigned long long) * 16));
    return 0;
}
#endif

int launch_mlp_wide_bwd(const MlpBwdArgs& a, hipStream_t s) {
    using N = NetNerf;
    auto k = mlp_wide_bwd_kernel<N>;
    const size_t lds = (size_t)(3 * N::HV + N::HW) * 4 + (size_t)WD_S * WD_SLOT + 8 * 4096 + 4 * 4096 + (size_t)WD_MT * WB_DPE_LD * 2 + 4096;
    if (a.scale == nullptr) return set_error("launch_mlp_wide_bwd: the fp16 chain needs its loss scale");
    static KernelOnce once;              // (per device: the attribute call costs more than the launch in the small configs)
    int dev = 0, n_cu = 0;
    if (int rc = current_device_cus(dev, n_cu)) return rc;
    if (int rc = kernel_lds_once(once, dev, reinterpret_cast<const void*>(k), lds)) return rc;
    const int tiles = (a.n_tiles * 128 + WD_MT - 1) / WD_MT;      // a.n_tiles counts 128-point tiles (point arrays are padded to 256)
    MlpBwdArgs b = a;
    b.n_tiles = tiles;
    const int grid = tiles < n_cu ? tiles : n_cu;                 // one workgroup per CU, tiles strided
    hipLaunchKernelGGL(k, dim3(grid), dim3(WD_NT), lds, s, b);
    LUSH_HIP(hipGetLastError());
    return 0;
}

}  // namespace lush
