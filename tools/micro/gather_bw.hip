// Developer micro-benchmark (needs a GPU; round 6, verdict item 2): what the weight-gradient kernel's LDS-DMA ring reads when its X rows
// are GATHERED through an index list instead of streamed -- the third form of the backward (stash-writing forward over ALL the points
// once, chain / weight gradients on the live list reading the stash rows of point live_idx[i]).  Rows of ROW bytes of an 8-GiB
// buffer, the list = a sorted random subset of the rows at the given share (the live list is in grid order), 256 workgroups of 8 waves,
// slots of 32 KiB, 4 slots (96 KiB in flight per CU, what dw_group_kernel keeps), one barrier per slot.  Indices come through scalar
// loads (lgkmcnt: a vector load would sit in front of the DMAs in vmcnt order).  Prints TB/s of rows delivered.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.hip -o build/gather_bw && ./build/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ROW, int STAGES>
__global__ __launch_bounds__(512) void gather_dma(const char* __restrict__ p, const int* __restrict__ idx, long long rows_per_wg, unsigned* __restrict__ out) {
    constexpr int SLOT = 32768, RPS = SLOT / ROW;              // rows per slot
    constexpr int PER_WAVE = SLOT / 8, INSTR = PER_WAVE / 1024;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const int* my = idx + (long long)blockIdx.x * rows_per_wg;
    const long long n_slots = rows_per_wg / RPS;
    auto issue = [&](long long s) {
        const unsigned dst = lds0 + (unsigned)(s % STAGES) * SLOT + w * PER_WAVE;
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            // this instruction's 1 KiB: bytes [w PER_WAVE + 1024 i, + 1024) of the slot = 1024 / ROW rows (or a piece of one)
            const int byte0 = w * PER_WAVE + i * 1024;
            unsigned long long row;
            unsigned col;
            if constexpr (ROW >= 1024) {
                row = (unsigned long long)my[s * RPS + byte0 / ROW];                 // wave-uniform: a scalar load
                col = (unsigned)(byte0 % ROW + lane * 16);
            } else {
                constexpr int LPR = ROW / 16;                                        // lanes per row
                const int r0 = my[s * RPS + byte0 / ROW], r1 = my[s * RPS + byte0 / ROW + 1];
                static_assert(ROW == 512, "two rows per instruction");
                row = (unsigned long long)(lane < LPR ? r0 : r1);
                col = (unsigned)((lane % LPR) * 16);
            }
            const char* src = p + row * ROW + col;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + i * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        }
    };
    for (long long s = 0; s < (long long)(STAGES - 1) && s < n_slots; ++s) issue(s);
    for (long long s = 0; s < n_slots; ++s) {
        if (s + STAGES - 1 <= n_slots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * INSTR < 63 ? (STAGES - 2) * INSTR : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + STAGES - 1 < n_slots) issue(s + STAGES - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x] == 0x7f && out) out[0] = 1;
}

// The same ring on a DENSE stream, in the four forms of its DMA instruction: scalar base + 32-bit per-lane offset (what the product's
// kernels issue) or a 64-bit per-lane address, each with the default or the non-temporal policy.
template <int FORM>      // 0: scalar base, default; 1: scalar base, nt; 2: per-lane address, default; 3: per-lane address, nt
__global__ __launch_bounds__(512) void stream_dma(const char* __restrict__ p, size_t bytes_per_wg, unsigned* __restrict__ out) {
    constexpr int SLOT = 32768, STAGES = 4, PER_WAVE = SLOT / 8, INSTR = PER_WAVE / 1024;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const char* q = p + (size_t)blockIdx.x * bytes_per_wg;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const size_t n_slots = bytes_per_wg / SLOT;
    auto issue = [&](size_t s) {
        const unsigned long long sv = (unsigned long long)(q + s * SLOT + (size_t)w * PER_WAVE);
        const unsigned s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sv), s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sv >> 32));
        const char* src = (const char*)(((unsigned long long)s_hi << 32) | (unsigned long long)s_lo);
        const unsigned dst = lds0 + (unsigned)(s % STAGES) * SLOT + w * PER_WAVE;
#pragma unroll
        for (int i = 0; i < INSTR; ++i) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + i * 1024);
            unsigned keep;
            const unsigned voff = (unsigned)(lane * 16 + i * 1024);
            if constexpr (FORM == 0)
                asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(m0v) : "memory");
            else if constexpr (FORM == 1)
                asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(m0v) : "memory");
            else if constexpr (FORM == 2)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src + voff), "s"(m0v) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src + voff), "s"(m0v) : "memory");
        }
    };
    for (size_t s = 0; s < (size_t)(STAGES - 1) && s < n_slots; ++s) issue(s);
    for (size_t s = 0; s < n_slots; ++s) {
        if (s + STAGES - 1 <= n_slots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * INSTR) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + STAGES - 1 < n_slots) issue(s + STAGES - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x] == 0x7f && out) out[0] = 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = (size_t)8 << 30;
    char* buf; unsigned* out; int* d_idx;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 12)); CK(hipMalloc(&d_idx, (bytes / 512) * 4));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](auto kern, int row, double share, const char* order) -> int {
        const long long n_rows = (long long)(bytes / row);
        std::vector<int> idx;
        unsigned s = 99;
        for (long long r = 0; r < n_rows; ++r) { s = s * 1664525u + 1013904223u; if ((s >> 8) / 16777216.0 < share) idx.push_back((int)r); }
        if (order[0] == 'r') { for (size_t i = idx.size(); i > 1; --i) { s = s * 1664525u + 1013904223u; std::swap(idx[i - 1], idx[(size_t)(s >> 4) % i]); } }
        const long long rps = 32768 / row, per_wg = (long long)(idx.size() / 256 / rps) * rps;
        CK(hipMemcpy(d_idx, idx.data(), (size_t)per_wg * 256 * 4, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 4 * 32768, 0, (const char*)buf, (const int*)d_idx, per_wg, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (r > 0 && ms < best) best = ms;
        }
        CK(hipDeviceSynchronize());
        printf("rows of %4d B, share %.2f of the rows, list %s: %.2f GB delivered in %.3f ms = %.2f TB/s\n", row, share, order, per_wg * 256.0 * row / 1e9, best, per_wg * 256.0 * row / best / 1e9);
        return 0;
    };
    for (double sh : {1.0, 0.6, 0.42, 0.3}) {
        if (run(gather_dma<512, 4>, 512, sh, "sorted (grid order)")) return 1;
        if (run(gather_dma<4096, 4>, 4096, sh, "sorted (grid order)")) return 1;
    }
    if (run(gather_dma<512, 4>, 512, 0.42, "random order")) return 1;
    // the dense stream in the four forms of the DMA instruction, on constant and on random bytes
    auto dense = [&](auto kern, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 4 * 32768, 0, (const char*)buf, bytes / 256, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (r > 0 && ms < best) best = ms;
        }
        printf("dense stream, %s: %.2f TB/s\n", name, bytes / best / 1e9);
    };
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {      // random bytes instead of the constant fill
            std::vector<unsigned> h((size_t)64 << 18);
            unsigned sd = 7;
            for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = sd; }
            for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy(buf + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            printf("-- random bytes --\n");
        }
        dense(stream_dma<0>, "scalar base + per-lane offset, default policy");
        dense(stream_dma<1>, "scalar base + per-lane offset, nt");
        dense(stream_dma<2>, "64-bit per-lane address, default policy");
        dense(stream_dma<3>, "64-bit per-lane address, nt");
    }
    return 0;
}
