// Developer probe (needs a GPU): the three software-visible hazards that round 3 met in hand-written `asm volatile`
// statements, measured in isolation.  hipcc pads none of them for an asm statement (it schedules the statement as one
// opaque instruction), so each probe puts producer, K wait states and consumer into ONE statement and reports for which K
// the consumer saw the NEW value in every lane of every wave.
//
//   P1  VALU writes an SGPR (v_readlane_b32: the compiler's scalar-spill reload; v_readfirstlane_b32 likewise)
//       -> a VMEM instruction reads that SGPR as its scalar base.           ISA table: 5 wait states.
//   P2  VALU writes VGPRs (v_mul_f32 / v_pk_mul_f32) -> v_mfma_f32_32x32x16_f16 reads them as SrcC.
//   P3  global_store_dwordx4 -> VALU overwrites the store's data VGPRs.      ISA table (gfx940+): 2 wait states.
//
//   P4  v_mfma_f32_32x32x16_f16 writes D  ->  a VALU instruction (a) reads a D register, (b) overwrites one.  ISA table: 8 passes
//       + 3 (hipcc pads 11 for its own code).  This is what round 4's audit found in the forward kernel: the register
//       allocator hands the dead tail of a partially-live accumulator block to asm outputs right behind the block's last MFMA.
// P1 cannot fault: the stale base is a valid buffer too (filled with 1.0, the new one with 2.0).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/hazards.hip -o build/hazards && ./build/hazards
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define NOP_0 ""
#define NOP_1 "s_nop 0\n\t"
#define NOP_2 "s_nop 1\n\t"
#define NOP_3 "s_nop 2\n\t"
#define NOP_4 "s_nop 3\n\t"
#define NOP_5 "s_nop 4\n\t"
#define NOP_6 "s_nop 5\n\t"
#define NOP_8 "s_nop 7\n\t"
#define NOP_10 "s_nop 9\n\t"
#define NOP_12 "s_nop 11\n\t"
#define NOP_16 "s_nop 15\n\t"

// ---------------------------------------------------------------------------------------------------------------- P1
#define P1_CASE(K)                                                                                                      \
    case K:                                                                                                             \
        asm volatile("s_mov_b64 s[20:21], %[a]\n\t"                                                                     \
                     "s_nop 7\n\t"                                                                                      \
                     "v_readlane_b32 s20, %[blo], 0\n\t"                                                                \
                     "v_readlane_b32 s21, %[bhi], 0\n\t" NOP_##K                                                        \
                     "global_load_dword %[r], %[off], s[20:21]\n\t"                                                     \
                     "s_waitcnt vmcnt(0)"                                                                               \
                     : [r] "=&v"(r)                                                                                     \
                     : [a] "s"(a), [blo] "v"(blo), [bhi] "v"(bhi), [off] "v"(off)                                       \
                     : "s20", "s21", "memory");                                                                         \
        break;

__global__ void p1_kernel(const float* A, const float* B, int k, int* bad) {
    const unsigned long long a = (unsigned long long)A, b = (unsigned long long)B;
    unsigned blo = (unsigned)b, bhi = (unsigned)(b >> 32);
    asm volatile("" : "+v"(blo), "+v"(bhi));
    const unsigned off = (threadIdx.x & 63) * 4;
    float r = 0.f;
    switch (k) {
        P1_CASE(0) P1_CASE(1) P1_CASE(2) P1_CASE(3) P1_CASE(4) P1_CASE(5) P1_CASE(6)
    }
    if (r != 2.0f) atomicAdd(bad, 1);
}

// ---------------------------------------------------------------------------------------------------------------- P2
// C = 1.0 in v[100:115]; v114 / v115 are multiplied by 3.0 by ONE producer; K wait states; D = 0 * 0 + C; out = D[14], D[15]
#define P2_BODY(PRODUCER, K)                                                                                            \
    asm volatile("v_mov_b32 v100, 1.0\n\tv_mov_b32 v101, 1.0\n\tv_mov_b32 v102, 1.0\n\tv_mov_b32 v103, 1.0\n\t"        \
                 "v_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 1.0\n\tv_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 1.0\n\t"        \
                 "v_mov_b32 v108, 1.0\n\tv_mov_b32 v109, 1.0\n\tv_mov_b32 v110, 1.0\n\tv_mov_b32 v111, 1.0\n\t"        \
                 "v_mov_b32 v112, 1.0\n\tv_mov_b32 v113, 1.0\n\tv_mov_b32 v114, 1.0\n\tv_mov_b32 v115, 1.0\n\t"        \
                 "v_mov_b32 v116, 0x40400000\n\tv_mov_b32 v117, 0x40400000\n\t"                                         \
                 "s_nop 7\n\t" PRODUCER NOP_##K                                                                         \
                 "v_mfma_f32_32x32x16_f16 v[100:115], %[z], %[z], v[100:115]\n\t"                                       \
                 "s_nop 15\n\ts_nop 15\n\t"                                                                             \
                 "v_mov_b32 %[o0], v114\n\tv_mov_b32 %[o1], v115"                                                       \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1)                                                                       \
                 : [z] "v"(z)                                                                                           \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",      \
                   "v112", "v113", "v114", "v115", "v116", "v117");
#define PK "v_pk_mul_f32 v[114:115], v[114:115], v[116:117]\n\t"
#define PL "v_mul_f32 v114, v114, v116\n\tv_mul_f32 v115, v115, v117\n\t"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void p2_kernel(int packed, int k, int* bad, unsigned* lanes) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    float o0 = 0.f, o1 = 0.f;
    if (packed) {
        switch (k) {
            case 0: P2_BODY(PK, 0) break; case 1: P2_BODY(PK, 1) break; case 2: P2_BODY(PK, 2) break; case 3: P2_BODY(PK, 3) break;
            case 4: P2_BODY(PK, 4) break; case 5: P2_BODY(PK, 5) break; case 6: P2_BODY(PK, 6) break; case 8: P2_BODY(PK, 8) break;
            case 12: P2_BODY(PK, 12) break;
        }
    } else {
        switch (k) {
            case 0: P2_BODY(PL, 0) break; case 1: P2_BODY(PL, 1) break; case 2: P2_BODY(PL, 2) break; case 3: P2_BODY(PL, 3) break;
            case 4: P2_BODY(PL, 4) break; case 5: P2_BODY(PL, 5) break; case 6: P2_BODY(PL, 6) break; case 8: P2_BODY(PL, 8) break;
            case 12: P2_BODY(PL, 12) break;
        }
    }
    if (o0 != 3.0f || o1 != 3.0f) {
        atomicAdd(bad, 1);
        atomicOr(&lanes[(threadIdx.x & 63) / 32], 1u << (threadIdx.x & 31));
    }
}

// ---------------------------------------------------------------------------------------------------------------- P3
#define P3_CASE(K)                                                                                                      \
    case K:                                                                                                             \
        asm volatile("v_mov_b32 v100, 5.0\n\tv_mov_b32 v101, 5.0\n\tv_mov_b32 v102, 5.0\n\tv_mov_b32 v103, 5.0\n\t"    \
                     "s_nop 7\n\t"                                                                                      \
                     "global_store_dwordx4 %[off], v[100:103], %[base]\n\t" NOP_##K                                     \
                     "v_mov_b32 v100, 0x41100000\n\tv_mov_b32 v101, 0x41100000\n\tv_mov_b32 v102, 0x41100000\n\t"       \
                     "v_mov_b32 v103, 0x41100000\n\t"                                                                   \
                     "s_waitcnt vmcnt(0)"                                                                               \
                     :                                                                                                  \
                     : [off] "v"(off), [base] "s"(out)                                                                  \
                     : "v100", "v101", "v102", "v103", "memory");                                                       \
        break;

__global__ void p3_kernel(float* out_all, int k) {
    float* out = out_all + (size_t)blockIdx.x * blockDim.x * 4;
    const unsigned off = threadIdx.x * 16;
    switch (k) {
        P3_CASE(0) P3_CASE(1) P3_CASE(2) P3_CASE(3) P3_CASE(4)
    }
}

// ---------------------------------------------------------------------------------------------------------------- P4
// C = 1.0, A = B = fp16 ones: D = 17.0 in every element.  (a) RAW: v_mov out, v101 after K wait states (1.0 = stale);
// (b) WAW: v_mov v101, 7.0 after K wait states, read back much later (17.0 = the MFMA's write landed on top of it).
#define P4_BODY(K, CONSUMER)                                                                                            \
    asm volatile("v_mov_b32 v100, 1.0\n\tv_mov_b32 v101, 1.0\n\tv_mov_b32 v102, 1.0\n\tv_mov_b32 v103, 1.0\n\t"        \
                 "v_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 1.0\n\tv_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 1.0\n\t"        \
                 "v_mov_b32 v108, 1.0\n\tv_mov_b32 v109, 1.0\n\tv_mov_b32 v110, 1.0\n\tv_mov_b32 v111, 1.0\n\t"        \
                 "v_mov_b32 v112, 1.0\n\tv_mov_b32 v113, 1.0\n\tv_mov_b32 v114, 1.0\n\tv_mov_b32 v115, 1.0\n\t"        \
                 "s_nop 7\n\t"                                                                                          \
                 "v_mfma_f32_32x32x16_f16 v[100:115], %[one], %[one], v[100:115]\n\t" NOP_##K CONSUMER                   \
                 "s_nop 15\n\ts_nop 15\n\t"                                                                             \
                 "v_mov_b32 %[o1], v101"                                                                                \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1)                                                                       \
                 : [one] "v"(one)                                                                                       \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",      \
                   "v112", "v113", "v114", "v115");
#define RD "v_mov_b32 %[o0], v101\n\t"
#define WR "v_mov_b32 v101, 0x40e00000\n\t"
__global__ void p4_kernel(int write, int k, int* bad) {
    const u32x4 one = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    float o0 = 0.f, o1 = 0.f;
    if (!write) {
        switch (k) {
            case 0: P4_BODY(0, RD) break; case 2: P4_BODY(2, RD) break; case 4: P4_BODY(4, RD) break; case 6: P4_BODY(6, RD) break;
            case 8: P4_BODY(8, RD) break; case 10: P4_BODY(10, RD) break; case 12: P4_BODY(12, RD) break; case 16: P4_BODY(16, RD) break;
        }
        if (o0 != 17.0f) atomicAdd(bad, 1);
    } else {
        switch (k) {
            case 0: P4_BODY(0, WR) break; case 2: P4_BODY(2, WR) break; case 4: P4_BODY(4, WR) break; case 6: P4_BODY(6, WR) break;
            case 8: P4_BODY(8, WR) break; case 10: P4_BODY(10, WR) break; case 12: P4_BODY(12, WR) break; case 16: P4_BODY(16, WR) break;
        }
        if (o1 != 7.0f) atomicAdd(bad, 1);
    }
}

int main() {
    const int blocks = 1024, threads = 256;
    float *A, *B, *out;
    int* bad;
    unsigned* lanes;
    CK(hipMalloc(&A, 256)); CK(hipMalloc(&B, 256)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&lanes, 8));
    CK(hipMalloc(&out, (size_t)blocks * threads * 16));
    std::vector<float> one(64, 1.0f), two(64, 2.0f);
    CK(hipMemcpy(A, one.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, two.data(), 256, hipMemcpyHostToDevice));
    printf("P1  v_readlane_b32 s -> global_load_dword ..., s[base]   (lanes that read through the STALE base, of %d)\n", blocks * threads);
    for (int k = 0; k <= 6; ++k) {
        CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(p1_kernel, dim3(blocks), dim3(threads), 0, 0, A, B, k, bad);
        int h = 0;
        CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("  wait states %d: %d stale\n", k, h);
    }
    for (int packed = 0; packed < 2; ++packed) {
        printf("P2  %s -> v_mfma_f32_32x32x16_f16 SrcC   (lanes with a stale accumulator, of %d; lane mask of the failures)\n",
               packed ? "v_pk_mul_f32" : "v_mul_f32 x2", blocks * threads);
        const int ks[] = {0, 1, 2, 3, 4, 5, 6, 8, 12};
        for (int k : ks) {
            CK(hipMemset(bad, 0, 4)); CK(hipMemset(lanes, 0, 8));
            hipLaunchKernelGGL(p2_kernel, dim3(blocks), dim3(threads), 0, 0, packed, k, bad, lanes);
            int h = 0; unsigned m[2] = {0, 0};
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(m, lanes, 8, hipMemcpyDeviceToHost));
            printf("  wait states %2d: %d stale, lanes 63..0 = %08x%08x\n", k, h, m[1], m[0]);
        }
    }
    for (int write = 0; write < 2; ++write) {
        printf("P4  v_mfma_f32_32x32x16_f16 D -> VALU %s of a D register   (lanes that saw the %s, of %d)\n", write ? "WRITE" : "READ",
               write ? "MFMA's write land on top of theirs" : "old value", blocks * threads);
        const int ks[] = {0, 2, 4, 6, 8, 10, 12, 16};
        for (int k : ks) {
            CK(hipMemset(bad, 0, 4));
            hipLaunchKernelGGL(p4_kernel, dim3(blocks), dim3(threads), 0, 0, write, k, bad);
            int h = 0;
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("  wait states %2d: %d wrong\n", k, h);
        }
    }
    printf("P3  global_store_dwordx4 -> v_mov_b32 of its data registers   (dwords that reached memory OVERWRITTEN, of %d)\n", blocks * threads * 4);
    std::vector<float> hout((size_t)blocks * threads * 4);
    for (int k = 0; k <= 4; ++k) {
        CK(hipMemset(out, 0, hout.size() * 4));
        hipLaunchKernelGGL(p3_kernel, dim3(blocks), dim3(threads), 0, 0, out, k);
        CK(hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost));
        long n9 = 0, nother = 0;
        for (float v : hout) { if (v == 9.0f) ++n9; else if (v != 5.0f) ++nother; }
        printf("  wait states %d: %ld overwritten, %ld neither\n", k, n9, nother);
    }
    return 0;
}
