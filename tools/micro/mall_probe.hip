// Developer micro-benchmark (needs a GPU): is a slice that one kernel has just WRITTEN read back by the next kernel faster than
// from HBM, i.e. from the 256 MiB Infinity Cache -- and up to which slice size?  This is the question behind cutting the
// backward's chain -> weight-gradient pair into slices: the chain kernel writes dZ (4 368 B per point,
// `nt` stores), the weight-gradient kernel streams dZ back (just written) together with the forward's stash X (written long ago:
// cold), through an LDS-DMA ring.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_probe.hip -o build/mall_probe && ./build/mall_probe
// Prints one table: for N = 32 .. 1024 MB: write rate; read-back rate right behind the write (same stream, next launch); read rate
// of the same bytes after 2 GiB of other traffic (cold); and the mixed case -- N MB just written + N MB cold in ONE read launch,
// which is what a weight-gradient slice would do.  256 workgroups (one per CU) and a 128-workgroup column for a CU split.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every workgroup writes its own contiguous slice, 16 bytes per lane and instruction
template <bool NT>
__global__ __launch_bounds__(256) void write_stream(uint4* __restrict__ p, size_t vec_per_wg, unsigned seed) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u* q = (v4u*)(p + (size_t)blockIdx.x * vec_per_wg);
    v4u v = {seed, threadIdx.x, blockIdx.x, 0x3c003c00u};
    for (size_t i = threadIdx.x; i < vec_per_wg; i += 256) {
        v.w += 1;
        if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
    }
}

// LDS-DMA ring as in the weight-gradient kernel: STAGES slots of SLOT bytes, 8 waves, one barrier per slot; a workgroup streams
// its slice of buffer A and (two = true) its slice of buffer B alternately, slot by slot.
template <int STAGES, int SLOT>
__global__ __launch_bounds__(512) void read_dma(const char* __restrict__ pa, const char* __restrict__ pb, size_t bytes_per_wg, int two,
                                                unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const char* qa = pa + (size_t)blockIdx.x * bytes_per_wg;
    const char* qb = pb + (size_t)blockIdx.x * bytes_per_wg;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PER_WAVE = SLOT / 8, INSTR = PER_WAVE / 1024;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const size_t n_half = bytes_per_wg / SLOT;                 // slots per buffer
    const size_t n_slots = two ? 2 * n_half : n_half;
    auto issue = [&](size_t s) {
        const char* base = two ? ((s & 1) ? qb : qa) + (s >> 1) * SLOT : qa + s * SLOT;
        const unsigned long long sv = (unsigned long long)(base + (size_t)w * PER_WAVE);
        const unsigned s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sv), s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sv >> 32));
        const char* src = (const char*)(((unsigned long long)s_hi << 32) | (unsigned long long)s_lo);
        const unsigned dst = lds0 + (unsigned)(s % STAGES) * SLOT + w * PER_WAVE;
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + i * 1024);
            unsigned keep;
            // (s_nop 4: the scalar base was just written by v_readfirstlane -- DESIGN.md section 4, "Hazards", rule R1)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((unsigned)(lane * 16 + i * 1024)), "s"(src), "s"(m0v) : "memory");
        }
    };
    for (size_t s = 0; s < (size_t)(STAGES - 1) && s < n_slots; ++s) issue(s);
    for (size_t s = 0; s < n_slots; ++s) {
        if (s + STAGES - 1 <= n_slots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * INSTR < 63 ? (STAGES - 2) * INSTR : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + STAGES - 1 < n_slots) issue(s + STAGES - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x] == 0x7f && out) out[0] = 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t MAXB = (size_t)1 << 30;                       // largest slice
    const size_t EVICT = (size_t)2 << 30;
    char *A, *B, *E; unsigned* out;
    CK(hipMalloc(&A, MAXB)); CK(hipMalloc(&B, MAXB)); CK(hipMalloc(&E, EVICT)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(A, 1, MAXB)); CK(hipMemset(B, 2, MAXB)); CK(hipMemset(E, 3, EVICT));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int STAGES = 4, SLOT = 32768;
    auto kread = read_dma<STAGES, SLOT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kread), hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * SLOT));
    auto ms_of = [&](auto launch) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto write = [&](char* p, size_t bytes, int wgs, bool nt, unsigned seed) {
        const size_t vec = bytes / 16 / wgs;
        if (nt) hipLaunchKernelGGL(write_stream<true>, dim3(wgs), dim3(256), 0, 0, (uint4*)p, vec, seed);
        else hipLaunchKernelGGL(write_stream<false>, dim3(wgs), dim3(256), 0, 0, (uint4*)p, vec, seed);
    };
    auto read = [&](const char* pa, const char* pb, size_t bytes, int wgs, int two) {
        hipLaunchKernelGGL(kread, dim3(wgs), dim3(512), STAGES * SLOT, 0, pa, pb, bytes / wgs / SLOT * SLOT, two, out);
    };
    auto evict = [&] { write(E, EVICT, 1024, true, 9); read(E, E, EVICT, 256, 0); };
    printf("MI355X Infinity-Cache probe: a slice written by one launch, read by the next through an LDS-DMA ring (4 x 32 KB per CU)\n");
    printf("rates in TB/s, median of 5; 'hot' = read right behind the write; 'cold' = after 2 GiB of other traffic; 'mixed' = N MB hot + N MB cold\n");
    printf("in one read launch (bytes of both / time)\n\n");
    for (int wgs : {256, 128}) {
        for (int nt = 1; nt >= 0; --nt) {
            printf("%d workgroups, %s stores\n", wgs, nt ? "nt" : "plain");
            printf("%8s %10s %10s %10s %10s %12s\n", "N (MB)", "write", "read hot", "read cold", "mixed", "hot / cold");
            for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024}) {
                const size_t bytes = mb << 20;
                std::vector<float> tw, th, tc, tm;
                for (int r = 0; r < 5; ++r) {
                    evict();
                    tw.push_back(ms_of([&] { write(A, bytes, wgs, nt, r); }));
                    th.push_back(ms_of([&] { read(A, A, bytes, wgs, 0); }));
                    evict();
                    tc.push_back(ms_of([&] { read(A, A, bytes, wgs, 0); }));
                    evict();
                    write(A, bytes, wgs, nt, r + 100);
                    tm.push_back(ms_of([&] { read(A, B, bytes, wgs, 1); }));          // A just written, B cold
                }
                const double w = bytes / med(tw) / 1e9, h = bytes / med(th) / 1e9, c = bytes / med(tc) / 1e9, m = 2.0 * bytes / med(tm) / 1e9;
                printf("%8zu %10.2f %10.2f %10.2f %10.2f %12.2f\n", mb, w, h, c, m, h / c);
            }
            printf("\n");
        }
    }
    // the overwrite question: does a slice buffer that is re-written every round (same addresses) keep its write traffic on chip?
    printf("re-writing ONE 128 MB buffer 16 times back to back (256 workgroups, nt): ");
    {
        const size_t bytes = (size_t)128 << 20;
        evict();
        float ms = ms_of([&] { for (int r = 0; r < 16; ++r) write(A, bytes, 256, true, r); });
        printf("%.2f TB/s; ", 16.0 * bytes / ms / 1e9);
        evict();
        ms = ms_of([&] { for (int r = 0; r < 16; ++r) write(A + (size_t)(r & 7) * bytes, bytes, 256, true, r); });
        printf("writing 8 different 128 MB buffers round robin: %.2f TB/s\n", 16.0 * bytes / ms / 1e9);
    }
    return 0;
}
