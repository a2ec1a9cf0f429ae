// lush-march: host-side helpers shared by the launch translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "lush_mlp.h"

namespace lush {

int set_error(const char* msg);                 // stores a thread-local message, returns -1
int set_hip_error(hipError_t e, const char* what, const char* file, int line);

#define LUSH_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) return ::lush::set_hip_error(_e, #expr, __FILE__, __LINE__); \
    } while (0)

// Launch set-up that is safe with several devices and several host threads in one process (the
// reference calls this path from one DataParallel worker thread per GPU): nothing here is keyed by "the first device
// that happened to call".  current_device_cus: CU count of the CALLING thread's current device (cached per device id);
// kernel_lds_once: hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) -- the call is idempotent,
// so two threads racing on the same device only repeat it.
struct KernelOnce { std::atomic<unsigned long long> done{0}; };
inline int current_device_cus(int& dev, int& n_cu) {
    static std::atomic<int> cus[64];
    LUSH_HIP(hipGetDevice(&dev));
    int v = dev >= 0 && dev < 64 ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (v <= 0) {
        LUSH_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        if (v <= 0) v = 256;
        if (dev >= 0 && dev < 64) cus[dev].store(v, std::memory_order_relaxed);
    }
    n_cu = v;
    return 0;
}
inline int kernel_lds_once(KernelOnce& once, int dev, const void* kernel, size_t lds) {
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 0 || dev >= 64 || !(once.done.load(std::memory_order_acquire) & bit)) {
        LUSH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}

// lush_abi.hip: the halves of lush_mlp_bwd for a caller whose lush_composite_bwd prepared the dstash header (lush_march_bwd)
// (prm / g are const lush_mlp_params* / const lush_mlp_grads*, stream a hipStream_t: passed untyped so that this header does not
// depend on include/lush_march.h)
int mlp_bwd_chain_prepared(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                           const void* packed_b, const void* prm, const float* draw, const void* stash,
                           void* dstash, float* dpts, int variant, void* stream, const int* live_idx = nullptr, const int* live_cnt = nullptr);
int mlp_bwd_weights_prepared(int net, int planes_f, int planes_b, int R, int S, const void* prm, const float* draw,
                             const void* stash, void* dstash, const void* g, int variant, void* stream, const int* live_cnt = nullptr);
bool mlp_live_kernels(int net, int planes_f, int planes_b, int variant);      // the kernels of this mode take live-point launches
bool mlp_dstash_header(int net, int planes_b, long long P, void* dstash, float** scale4, float** zero_buf, long long* zero_n);

// lush_mlp.hip
size_t mlp_fwd_lds_bytes(int hw, int ns, int mt);
size_t mlp_bwd_lds_bytes(int hw, int ns, int mt, int nthreads);
int mlp_fwd_tile(int ns);
int mlp_bwd_tile(int ns);
int launch_mlp_fwd(int net, int ns, const MlpFwdArgs& a, int grid, hipStream_t s);
int launch_mlp_bwd(int net, int ns, const MlpBwdArgs& a, int grid, hipStream_t s);
// lush_mlp_chain.hip
bool mlp_fwd_chain_enabled(int planes);
int launch_mlp_chain_fwd(int net, int planes, const MlpFwdArgs& a, int variant, hipStream_t s);
bool mlp_bwd_chain_enabled(int planes);
int launch_mlp_chain_bwd(int net, int planes, const MlpBwdArgs& a, int variant, hipStream_t s);
// lush_mlp_wide.hip
size_t mlp_wide_fwd_lds_bytes();
int launch_mlp_wide_fwd(const MlpFwdArgs& a, hipStream_t s);
// lush_mlp_wide_bwd.hip
int launch_mlp_wide_bwd(const MlpBwdArgs& a, hipStream_t s);
int launch_pack(int ns, const PackTable& t, int total_blocks, void* dst, hipStream_t s);
int launch_pack_plan(const void* plan, int blocks, hipStream_t s, float* zero_buf, long long zero_n);
int launch_pack_f32(int net, int ns, const MlpParams& prm, void* packed, hipStream_t s);
int launch_dw(int ns, const DwArgs& a, int splits, hipStream_t s);
int launch_dw_group(const DwGroup& g, int splits, int ns, bool x_f16, bool z_f16, hipStream_t s);
int launch_feat_factor(const FeatFactorArgs& a, hipStream_t s);
int launch_grad_scale(const float* draw, long long n, float* scale, float* zero_buf, long long zero_n, hipStream_t s);
int launch_head_dw(int ns, bool x_f16, const float* draw, long long P, const __bf16* hv, long long plane_hv, int HV,
                   const __bf16* hl, long long plane_h, int HW, float* dw_rgb, float* db_rgb, float* dw_alpha,
                   float* db_alpha, hipStream_t s, const int* live_cnt = nullptr);

}  // namespace lush
