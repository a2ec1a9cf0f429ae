// Developer micro-test (needs a GPU): accuracy of a v_sin_f32 / v_cos_f32 based positional encoding against float64.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sincos_hw.hip -o ab/sincos_hw && ./ab/sincos_hw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
// revolutions of 2^k * x, accurate to ~1e-9: x / (2 pi) as a double-float (hi + lo), the power of two and fract are exact
__device__ __forceinline__ void sincos_rev(float x, int k, float* sn, float* cs) {
    const float C_HI = 0.15915494309189535f, C_LO = -6.5720892e-09f * 0.f + (float)(0.15915494309189533576888 - (double)0.15915494309189535f);
    const float hi = x * C_HI;
    const float lo = __builtin_fmaf(x, C_HI, -hi) + x * C_LO;
    const float s = (float)(1 << k);
    float r = __builtin_amdgcn_fractf(hi * s) + lo * s;
    *sn = __builtin_amdgcn_sinf(r);
    *cs = __builtin_amdgcn_cosf(r);
}
__global__ void k(const float* x, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int f = 0; f < 10; ++f) sincos_rev(x[i], f, &out[(i * 10 + f) * 2], &out[(i * 10 + f) * 2 + 1]);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = -2.0f + 4.0f * (float)((i * 2654435761u) >> 8) / (float)(1 << 24);
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 80);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<float> ho(n * 20);
    hipMemcpy(ho.data(), dout, n * 80, hipMemcpyDeviceToHost);
    for (int f = 0; f < 10; ++f) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double a = (double)hx[i] * (double)(1 << f);
            worst = fmax(worst, fabs((double)ho[(i * 10 + f) * 2] - sin(a)));
            worst = fmax(worst, fabs((double)ho[(i * 10 + f) * 2 + 1] - cos(a)));
        }
        printf("freq 2^%d: max abs error vs float64 %.3e\n", f, worst);
    }
    return 0;
}
