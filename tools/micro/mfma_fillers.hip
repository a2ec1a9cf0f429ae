// Developer micro-benchmark (needs a GPU): which filler instructions hide behind v_mfma_f32_32x32x16_f16 when ONE wave per
// SIMD issues them between its MFMAs?  Prints cycles per MFMA for K fillers of each kind per MFMA gap.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fillers.hip -o ab/mfma_fillers && ./ab/mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP4(x) x x x x
template <int KIND, int K>
__device__ __forceinline__ void filler(unsigned (&A)[4], unsigned& b, float (&F)[4], float& g, unsigned& acc_dummy) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        unsigned& a = A[i % 4];      // four independent dependency chains
        float& f = F[i % 4];
        if constexpr (KIND == 0) { }
        else if constexpr (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f) : "v"(g));
        else if constexpr (KIND == 2) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 3) asm volatile("v_pk_sub_u16 %0, 1, %0 op_sel_hi:[0,1] clamp" : "+v"(a));
        else if constexpr (KIND == 4) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(f), "v"(g));
        else if constexpr (KIND == 6) asm volatile("v_med3_i32 %0, %0, 0, 1" : "+v"(a));
        else if constexpr (KIND == 7) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 8) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc_dummy) : "v"(a));
        else if constexpr (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a) : "a"(acc_dummy));
        else if constexpr (KIND == 10) asm volatile("s_nop 0");
        else if constexpr (KIND == 11) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g));
        else if constexpr (KIND == 13) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 14) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f), "v"(g) : "vcc");
        else if constexpr (KIND == 15) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(a) : "v"(f), "v"(g));
        else if constexpr (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 17) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a));
        else if constexpr (KIND == 18) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 19) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
        else if constexpr (KIND == 20) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(&F[0])) : "v"(b));
        else if constexpr (KIND == 21) {      // a realistic conversion mix: cvt, clamp, flag, merge
            if (i % 4 == 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(f), "v"(g));
            else if (i % 4 == 1) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(A[(i - 1) % 4]) : "v"(b));
            else if (i % 4 == 2) asm volatile("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(a) : "v"(A[(i - 2) % 4]));
            else asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(A[3]) : "v"(A[(i - 1) % 4]));
        }
    }
}

template <int KIND, int K>
__global__ __launch_bounds__(256) void kern(const f16x8* in, float* out, unsigned long long* cyc, int iters) {
    f16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 256];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned ua[4] = {threadIdx.x * 3 + 1, threadIdx.x * 5 + 2, threadIdx.x * 7 + 3, threadIdx.x * 11 + 4}, ub = threadIdx.x + 7, ad = 0;
    float f[4] = {threadIdx.x * 0.5f, threadIdx.x * 0.25f, threadIdx.x * 0.125f, threadIdx.x * 2.f}, g = 1.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        filler<KIND, K>(ua, ub, f, g, ad);
        __builtin_amdgcn_sched_barrier(0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
        filler<KIND, K>(ua, ub, f, g, ad);
        __builtin_amdgcn_sched_barrier(0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c2, 0, 0, 0);
        filler<KIND, K>(ua, ub, f, g, ad);
        __builtin_amdgcn_sched_barrier(0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c3, 0, 0, 0);
        filler<KIND, K>(ua, ub, f, g, ad);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + f[0] + f[1] + f[2] + f[3] + (float)(ua[0] + ua[1] + ua[2] + ua[3]) + (float)ad;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int K>
double run(const f16x8* in, float* out, unsigned long long* cyc, int grid) {
    const int iters = 4000;
    hipLaunchKernelGGL((kern<KIND, K>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((kern<KIND, K>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return (double)c / (4.0 * iters);
}

template <int KIND>
void row(const char* name, const f16x8* in, float* out, unsigned long long* cyc, int grid) {
    printf("%-22s K=0 %5.1f  K=2 %5.1f  K=3 %5.1f  K=4 %5.1f  K=5 %5.1f  K=6 %5.1f  K=8 %5.1f  (cycles per MFMA)\n", name, run<0, 0>(in, out, cyc, grid),
           run<KIND, 2>(in, out, cyc, grid), run<KIND, 3>(in, out, cyc, grid), run<KIND, 4>(in, out, cyc, grid), run<KIND, 5>(in, out, cyc, grid),
           run<KIND, 6>(in, out, cyc, grid), run<KIND, 8>(in, out, cyc, grid));
}

int main() {
    f16x8* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.001f * (float)((i * 2654435761u) % 1000) - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int grid : {256}) {
        printf("== grid %d (one 256-thread workgroup per CU: one wave per SIMD)\n", grid);
        row<1>("v_max_f32", in, out, cyc, grid);
        row<12>("v_add_f32", in, out, cyc, grid);
        row<11>("v_or_b32", in, out, cyc, grid);
        row<18>("v_max_i32", in, out, cyc, grid);
        row<2>("v_pk_max_i16", in, out, cyc, grid);
        row<3>("v_pk_sub_u16 clamp", in, out, cyc, grid);
        row<13>("v_pk_add_f16", in, out, cyc, grid);
        row<4>("v_dot2_u32_u16", in, out, cyc, grid);
        row<5>("v_cvt_pk_f16_f32", in, out, cyc, grid);
        row<15>("v_cvt_pkrtz_f16_f32", in, out, cyc, grid);
        row<6>("v_med3_i32", in, out, cyc, grid);
        row<7>("v_lshl_or_b32", in, out, cyc, grid);
        row<19>("v_and_or_b32", in, out, cyc, grid);
        row<16>("v_perm_b32", in, out, cyc, grid);
        row<17>("v_bfe_u32", in, out, cyc, grid);
        row<14>("v_cmp_gt_f32", in, out, cyc, grid);
        row<8>("v_accvgpr_write_b32", in, out, cyc, grid);
        row<9>("v_accvgpr_read_b32", in, out, cyc, grid);
        row<10>("s_nop 0", in, out, cyc, grid);
        row<21>("cvt/max/flag/merge mix", in, out, cyc, grid);
    }
    return 0;
}
