// Developer micro-benchmark (needs a GPU): fp16 MFMA shape under load -- v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 in the
// forward's skeleton (weights = A operand re-read from LDS by ds_read_b128, every fragment feeding 64 points; activations = B operand in
// registers; 64 accumulator registers), random data, one and two waves per SIMD.  Prints wall TFLOP/s, cycles per 32 KFLOP and the
// in-kernel clock (s_memtime / s_memrealtime).  MI355X_MICROARCH.md 'DVFS give-back' item 7 quotes 1.12-1.15 x for the bf16 forms.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o build/mfma_shape && ./build/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// One "position" = 8 weight fragments (8 KiB of LDS) against the wave's 64 points: 16 MFMAs 32x32x16 or 32 MFMAs 16x16x32.
template <int SHAPE, int NT>
__global__ __launch_bounds__(NT) void kern(const f16x8* w, const f16x8* x, float* out, unsigned long long* st, int iters) {
    __shared__ f16x8 frag[64 * 64];        // 64 KiB of fragments, walked cyclically
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 64; i += NT) frag[i] = w[i];
    f16x8 b[16];                           // the B operand: 64 registers of activations (a quarter of a layer's operand)
    for (int i = 0; i < 16; ++i) b[i] = x[(blockIdx.x * NT + tid) * 16 + i];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    // fragments of position it + 1 are read while position it's MFMAs run (two register sets, the loop runs two positions per trip)
    f16x8 fa[8], fb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) fa[u] = frag[u * 64 + lane];
    auto body = [&](auto& acc, const f16x8 (&cur)[8], f16x8 (&nxt)[8], int it, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            nxt[u] = frag[(((it + 1) * 8 + u) & 63) * 64 + lane];
            if constexpr (SHAPE == 32) {       // unit = (k-block u / 2, row block u % 2): one fragment, two column blocks of 32
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c][u % 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[u], b[(u / 2) * 2 + c + 8 * half], acc[c][u % 2], 0, 0, 0);
            } else {                           // unit = (k-block (32 wide) u / 4, row block u % 4): one fragment, four column blocks of 16
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c][u % 4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[u], b[(u / 4) * 4 + c + 8 * half], acc[c][u % 4], 0, 0, 0);
            }
        }
    };
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; it += 2) { body(acc, fa, fb, it, 0); body(acc, fb, fa, it + 1, 1); }
        for (int i = 0; i < 16; ++i) s += acc[0][0][i] + acc[0][1][i] + acc[1][0][i] + acc[1][1][i];
    } else {
        f32x4 acc[4][4] = {};                  // [column block of 16][row block of 16]
        for (int it = 0; it < iters; it += 2) { body(acc, fa, fb, it, 0); body(acc, fb, fa, it + 1, 1); }
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) for (int i = 0; i < 4; ++i) s += acc[c][r][i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * NT + tid] = s;
    if (tid == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int NT>
void run(const char* name, const f16x8* w, const f16x8* x, float* out, unsigned long long* st) {
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL((kern<SHAPE, NT>), dim3(256), dim3(NT), 0, 0, w, x, out, st, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; tick += (double)h[2 * i + 1]; }
    const double fl = 256.0 * (NT / 64) * iters * 16.0 * 32768.0;       // a position = 16 x 32 KFLOP per wave
    printf("%-44s %7.3f ms  %7.0f TFLOP/s  %5.1f cycles per 32 KFLOP and SIMD  clock %4.0f MHz\n", name, best, fl / best / 1e9, cyc / 256 / iters / 16.0 / (NT / 256), cyc / tick * 100.0);
}

int main() {
    f16x8 *w, *x; float* out; unsigned long long* st;
    hipMalloc(&w, 64 * 64 * 16); hipMalloc(&x, 256 * 512 * 16 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 512 * 8);
    std::vector<_Float16> h((size_t)256 * 512 * 16 * 8);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.0f - 0.5f); }
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data() + 4096, 64 * 64 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 256>("32x32x16 f16, one wave per SIMD", w, x, out, st);
        run<16, 256>("16x16x32 f16, one wave per SIMD", w, x, out, st);
        run<32, 512>("32x32x16 f16, two waves per SIMD", w, x, out, st);
        run<16, 512>("16x16x32 f16, two waves per SIMD", w, x, out, st);
    }
    hipMemset(x, 0, 256 * 512 * 16 * 16); hipMemset(w, 0, 64 * 64 * 16);
    run<32, 256>("32x32x16 f16, one wave per SIMD, ZEROS", w, x, out, st);
    run<16, 256>("16x16x32 f16, one wave per SIMD, ZEROS", w, x, out, st);
    return 0;
}
