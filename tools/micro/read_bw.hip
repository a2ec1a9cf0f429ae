// Developer micro-benchmark (needs a GPU): what a READ-ONLY stream reaches on this box, as a function of the bytes in flight per CU,
// in the two forms the kernels use -- 16-byte loads into registers, and LDS-DMA (global_load_lds_dwordx4) into a ring.  The
// weight-gradient kernel (dw_group_kernel) is such a stream: 108 KB in flight per CU through LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/read_bw.hip -o build/read_bw && ./build/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every workgroup streams its own contiguous slice; U independent 16-byte loads per thread in flight
template <int U>
__global__ __launch_bounds__(256) void read_regs(const uint4* __restrict__ p, size_t vec_per_wg, unsigned* __restrict__ out) {
    const uint4* q = p + (size_t)blockIdx.x * vec_per_wg;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * 256 < vec_per_wg; i += (size_t)U * 256) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[i + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// LDS-DMA ring: STAGES slots of SLOT bytes; each of the 8 waves moves 1/8 of a slot with 16-byte pieces (64 lanes x 16 B = 1 KiB per
// instruction); one barrier per slot, like the kernels' rings.  Nothing reads the LDS: this is the transport alone.
template <int STAGES, int SLOT, bool BARRIER = true>
__global__ __launch_bounds__(512) void read_dma(const char* __restrict__ p, size_t bytes_per_wg, unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const char* q = p + (size_t)blockIdx.x * bytes_per_wg;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PER_WAVE = SLOT / 8, INSTR = PER_WAVE / 1024;        // 1-KiB pieces per wave and slot
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const size_t n_slots = bytes_per_wg / SLOT;
    auto issue = [&](size_t s) {
        const unsigned long long sv = (unsigned long long)(q + s * SLOT + (size_t)w * PER_WAVE);      // wave-uniform: into an SGPR pair
        const unsigned s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sv), s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sv >> 32));
        const char* src = (const char*)(((unsigned long long)s_hi << 32) | (unsigned long long)s_lo);      // (unsigned: readfirstlane returns int)
        const unsigned dst = lds0 + (unsigned)(s % STAGES) * SLOT + w * PER_WAVE;
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + i * 1024);
            unsigned keep;
            // (s_nop 4 first: the scalar base was just written by v_readfirstlane, and a VMEM instruction may read a VALU-written SGPR only
            // five wait states later -- DESIGN.md section 4, "Hazards", rule R1)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((unsigned)(lane * 16 + i * 1024)), "s"(src), "s"(m0v) : "memory");
        }
    };
    for (size_t s = 0; s < (size_t)(STAGES - 1) && s < n_slots; ++s) issue(s);
    for (size_t s = 0; s < n_slots; ++s) {
        // all but the STAGES - 2 younger slots' pieces of this wave have landed
        if (s + STAGES - 1 <= n_slots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * INSTR < 63 ? (STAGES - 2) * INSTR : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __syncthreads();
        if (s + STAGES - 1 < n_slots) issue(s + STAGES - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x] == 0x7f && out) out[0] = 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = (size_t)8 << 30;
    char* buf; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
        return best;
    };
    for (int wgs : {256, 512, 1024, 2048}) {
        const size_t vec_per_wg = bytes / 16 / wgs;
        float t1 = time([&] { hipLaunchKernelGGL(read_regs<1>, dim3(wgs), dim3(256), 0, 0, (const uint4*)buf, vec_per_wg, out); });
        float t4 = time([&] { hipLaunchKernelGGL(read_regs<4>, dim3(wgs), dim3(256), 0, 0, (const uint4*)buf, vec_per_wg, out); });
        float t8 = time([&] { hipLaunchKernelGGL(read_regs<8>, dim3(wgs), dim3(256), 0, 0, (const uint4*)buf, vec_per_wg, out); });
        float t16 = time([&] { hipLaunchKernelGGL(read_regs<16>, dim3(wgs), dim3(256), 0, 0, (const uint4*)buf, vec_per_wg, out); });
        printf("registers, %4d workgroups of 256: 4 / 16 / 32 / 64 KB in flight each: %.2f %.2f %.2f %.2f TB/s\n", wgs, bytes / t1 / 1e9, bytes / t4 / 1e9,
               bytes / t8 / 1e9, bytes / t16 / 1e9);
    }
    {
        const int wgs = 256;
        const size_t per = bytes / wgs;
        auto run = [&](auto k, int stages, int slot, const char* note = "") {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, stages * slot);
            float t = time([&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(512), stages * slot, 0, (const char*)buf, per, out); });
            printf("LDS-DMA, 256 workgroups of 512, %d slots of %d KB (%d KB in flight)%s: %.2f TB/s\n", stages, slot >> 10, (stages - 1) * slot >> 10, note, bytes / t / 1e9);
        };
        run(read_dma<3, 32768>, 3, 32768);
        run(read_dma<4, 32768>, 4, 32768);
        run(read_dma<5, 32768>, 5, 32768);
        run(read_dma<4, 16384>, 4, 16384);
        run(read_dma<8, 16384>, 8, 16384);
        run(read_dma<9, 16384>, 9, 16384);
        run(read_dma<16, 8192>, 16, 8192);
        run(read_dma<4, 32768, false>, 4, 32768, ", no barrier per slot");
        run(read_dma<8, 16384, false>, 8, 16384, ", no barrier per slot");
    }
    return 0;
}
