// lush-march: ray-level kernels (everything around the MLPs) + their C ABI.
// One wavefront (64 lanes) owns one ray: 64 coarse samples map 1:1 onto lanes and
// scans/reductions along the ray are wave-level shuffles.
#include "lush_common.h"
#include "lush_host.h"
#include <type_traits>
#include "../../include/lush_march.h"

#include <cmath>
#include <string>

namespace lush {

static thread_local std::string g_err;
int set_error(const char* msg) { g_err = msg; return -1; }
int set_hip_error(hipError_t e, const char* what, const char* file, int line) {
    g_err = std::string(hipGetErrorString(e)) + " in " + what + " at " + file + ":" + std::to_string(line);
    return -2;
}

constexpr int RAYS_PER_BLOCK = 4;   // 4 waves per workgroup, one ray each

// torch.linspace(0, 1, n)[i] as ATen computes it in fp32
// (aten/src/ATen/native/RangeFactories: step=(end-start)/(n-1); first half counts
// up from start, second half counts down from end).
__device__ __forceinline__ float linspace01(int i, int n) {
    if (n == 1) return 0.f;
    const float step = 1.0f / (float)(n - 1);
    // the count-down half is one fused multiply-add in ATen's vectorised kernel (verified bit-exact
    // against torch.linspace for n = 2..256)
    return i < n / 2 ? __fmul_rn(step, (float)i) : __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
}

__device__ __forceinline__ float zgrid_at(float near, float far, int i, int S, int lindisp) {
    const float t = linspace01(i, S);
    if (!lindisp) return __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, t)), __fmul_rn(far, t));
    return 1.f / __fadd_rn(__fmul_rn(1.f / near, __fsub_rn(1.f, t)), __fmul_rn(1.f / far, t));
}

// --------------------------------------------------------------------- z grid
__global__ void zgrid_kernel(const float* __restrict__ rays, int R, int S, int lindisp,
                             const float* __restrict__ t_rand, float* __restrict__ z) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * S) return;
    const int ray = (int)(i / S), s = (int)(i % S);
    const float near = rays[ray * 11 + 6], far = rays[ray * 11 + 7];
    const float zc = zgrid_at(near, far, s, S, lindisp);
    if (t_rand == nullptr) { z[i] = zc; return; }
    const float zl = s > 0 ? zgrid_at(near, far, s - 1, S, lindisp) : zc;
    const float zr = s < S - 1 ? zgrid_at(near, far, s + 1, S, lindisp) : zc;
    const float lower = s > 0 ? __fmul_rn(.5f, __fadd_rn(zc, zl)) : zc;
    const float upper = s < S - 1 ? __fmul_rn(.5f, __fadd_rn(zr, zc)) : zc;
    z[i] = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[i]));
}

__global__ void zfixed_kernel(const float* __restrict__ rays, int R, int S, int index, int lindisp,
                              float* __restrict__ z) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray < R) z[ray] = zgrid_at(rays[ray * 11 + 6], rays[ray * 11 + 7], index, S, lindisp);
}

// ---------------------------------------------------------------- compositing
// Lane l owns samples l*SPL .. l*SPL+SPL-1.
struct CompIn {
    const float *raw, *z, *rays, *noise;
    int R, S;
    float noise_std, near_mask;
    int white_bkgd;
};

template <int SPL>
__device__ __forceinline__ void comp_load(const CompIn& c, int ray, int lane, float (&alpha)[SPL], float (&dist)[SPL],
                                          float (&dens)[SPL], float (&gate)[SPL], float (&zz)[SPL],
                                          float (&rgbv)[SPL][3], float& norm, bool* bad_raw = nullptr) {
    const float* rr = c.rays + (long long)ray * 11;
    norm = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        const int j = lane * SPL + e;
        alpha[e] = 0.f; dist[e] = 0.f; dens[e] = 0.f; gate[e] = 0.f; zz[e] = 0.f;
        rgbv[e][0] = rgbv[e][1] = rgbv[e][2] = 0.f;
        if (j < c.S) {
            const long long p = (long long)ray * c.S + j;
            const float4 rw = *reinterpret_cast<const float4*>(c.raw + p * 4);
            if (bad_raw)     // NaN or Inf in the network output (models/lushnerf.py:474-478 checks ret['raw'])
                *bad_raw |= !(fabsf(rw.x) <= 3.402823466e38f) || !(fabsf(rw.y) <= 3.402823466e38f) ||
                            !(fabsf(rw.z) <= 3.402823466e38f) || !(fabsf(rw.w) <= 3.402823466e38f);
            zz[e] = c.z[p];
            rgbv[e][0] = 1.f / (1.f + expf(-rw.x));
            rgbv[e][1] = 1.f / (1.f + expf(-rw.y));
            rgbv[e][2] = 1.f / (1.f + expf(-rw.z));
            if (j < c.S - 1) {
                const float znext = c.z[p + 1];
                dist[e] = (znext - zz[e]) * norm;
                float pre = rw.w;
                if (c.noise != nullptr && c.noise_std > 0.f)
                    pre += c.noise[(long long)ray * (c.S - 1) + j] * c.noise_std;
                float g = pre > 0.f ? 1.f : 0.f;
                float d = fmaxf(pre, 0.f);
                if (c.near_mask >= 0.f && !(znext > c.near_mask)) { d = 0.f; g = 0.f; }
                dens[e] = d; gate[e] = g;
                alpha[e] = 1.f - expf(-d * dist[e]);
            } else {
                alpha[e] = 1.f;   // last sample: models/lushnerf.py:338
            }
        }
    }
}

// transmittance T_j = prod_{k<j} (1 - alpha_k)
template <int SPL>
__device__ __forceinline__ void comp_trans(const float (&alpha)[SPL], int lane, float (&T)[SPL]) {
    float loc = 1.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) { T[e] = loc; loc *= (1.f - alpha[e]); }
    float incl = wave_incl_prod(loc, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) T[e] *= excl;
}

template <int SPL>
__global__ __launch_bounds__(RAYS_PER_BLOCK * 64) void composite_fwd_kernel(CompIn c, float* __restrict__ rgb,
        float* __restrict__ depth, float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ density,
        int* __restrict__ flags, int flag_shift) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= c.R) return;
    float alpha[SPL], dist[SPL], dens[SPL], gate[SPL], zz[SPL], col[SPL][3], T[SPL], norm;
    bool bad_dens = false, bad_raw = false;
    comp_load<SPL>(c, ray, lane, alpha, dist, dens, gate, zz, col, norm, flags != nullptr ? &bad_raw : nullptr);
    comp_trans<SPL>(alpha, lane, T);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        const int j = lane * SPL + e;
        const float wgt = alpha[e] * T[e];
        if (j < c.S) {
            if (weights) weights[(long long)ray * c.S + j] = wgt;
            if (density && j < c.S - 1) density[(long long)ray * (c.S - 1) + j] = dens[e];
            if (j < c.S - 1) bad_dens |= !(fabsf(dens[e]) <= 3.402823466e38f);
        }
        s0 += wgt * col[e][0]; s1 += wgt * col[e][1]; s2 += wgt * col[e][2];
        sd += wgt * zz[e]; sa += wgt;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); sd = wave_sum(sd); sa = wave_sum(sa);
    if (flags != nullptr) {   // wave-uniform branch; the ballots run with all lanes active
        const bool any_dens = __ballot(bad_dens) != 0ull, any_raw = __ballot(bad_raw) != 0ull;
        if (lane == 0) {
            if (c.white_bkgd) { s0 += 1.f - sa; s1 += 1.f - sa; s2 += 1.f - sa; }
            auto bad = [](float v) { return !(fabsf(v) <= 3.402823466e38f); };      // NaN or Inf
            const int w = (bad(s0) || bad(s1) || bad(s2) ? LUSH_FAULT_RGB : 0) | (bad(sd) ? LUSH_FAULT_DEPTH : 0) |
                          (bad(sa) ? LUSH_FAULT_ACC : 0) | (any_dens ? LUSH_FAULT_DENSITY : 0) | (any_raw ? LUSH_FAULT_RAW : 0);
            if (w) atomicOr(flags, w << flag_shift);
            rgb[ray * 3 + 0] = s0; rgb[ray * 3 + 1] = s1; rgb[ray * 3 + 2] = s2;
            depth[ray] = sd; acc[ray] = sa;
        }
        return;
    }
    if (lane == 0) {
        if (c.white_bkgd) { s0 += 1.f - sa; s1 += 1.f - sa; s2 += 1.f - sa; }
        rgb[ray * 3 + 0] = s0; rgb[ray * 3 + 1] = s1; rgb[ray * 3 + 2] = s2;
        depth[ray] = sd; acc[ray] = sa;
    }
}

// What the backward of a pass needs besides d_raw, folded into this kernel so that it costs no launch of its own (round 4):
//   block_max  per-workgroup max |d_raw| (one float per workgroup of this launch, plain stores), or NULL: loss_scale_kernel turns
//            them into the fp16 gradient chain's loss scale.  (A first version ended in the "last workgroup to finish" pattern
//            of grad_scale_kernel: 2 x 2048 atomics on one word took this kernel from 27 to 127 us.)
//   zero_buf a scratch the weight-gradient launch accumulates into (the feature-factor block), zeroed here, or NULL;
//   init_drays: drays is written whole (zeros outside columns 3..5) instead of added to -- the first pass of a march.
struct CompBwdExtra { float* block_max; float* zero_buf; long long zero_n; int init_drays; };

template <int SPL>
__global__ __launch_bounds__(RAYS_PER_BLOCK * 64) void composite_bwd_kernel(CompIn c, const float* __restrict__ g_rgb,
        const float* __restrict__ g_depth, const float* __restrict__ g_acc, float* __restrict__ draw,
        float* __restrict__ drays, CompBwdExtra X) {
    const int lane = threadIdx.x & 63;
    if (X.zero_buf != nullptr)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < X.zero_n; i += (long long)gridDim.x * blockDim.x) X.zero_buf[i] = 0.f;
    float vmax = 0.f;
  for (int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6); ray < c.R; ray += gridDim.x * RAYS_PER_BLOCK) {
    float alpha[SPL], dist[SPL], dens[SPL], gate[SPL], zz[SPL], col[SPL][3], T[SPL], norm;
    comp_load<SPL>(c, ray, lane, alpha, dist, dens, gate, zz, col, norm);
    comp_trans<SPL>(alpha, lane, T);
    float gr[3] = {0.f, 0.f, 0.f};
    if (g_rgb) { gr[0] = g_rgb[ray * 3]; gr[1] = g_rgb[ray * 3 + 1]; gr[2] = g_rgb[ray * 3 + 2]; }
    const float gd = g_depth ? g_depth[ray] : 0.f;
    float ga = g_acc ? g_acc[ray] : 0.f;
    if (c.white_bkgd) ga -= gr[0] + gr[1] + gr[2];
    // G_j = dL/dw_j ; S_j = sum_{k>j} G_k alpha_k prod_{j<m<k}(1-alpha_m): reverse scan of the
    // affine maps f_k(s) = G_k alpha_k + (1-alpha_k) s, composed (a1,b1)o(a2,b2) = (a1 a2, a1 b2 + b1).
    float G[SPL];
#pragma unroll
    for (int e = 0; e < SPL; ++e) G[e] = gr[0] * col[e][0] + gr[1] * col[e][1] + gr[2] * col[e][2] + gd * zz[e] + ga;
    // lane aggregate F = f_{first} o ... o f_{last} of this lane's samples
    float Fa = 1.f, Fb = 0.f;
#pragma unroll
    for (int e = SPL - 1; e >= 0; --e) {   // F <- f_e o F
        Fb = G[e] * alpha[e] + (1.f - alpha[e]) * Fb;
        Fa = (1.f - alpha[e]) * Fa;
    }
    // suffix composition over lanes: Suf_l = F_l o F_{l+1} o ... o F_63
    float Sa = Fa, Sb = Fb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float ta = __shfl_down(Sa, o, 64), tb = __shfl_down(Sb, o, 64);
        if (lane + o < 64) { Sb = Sa * tb + Sb; Sa = Sa * ta; }
    }
    float tail = __shfl_down(Sb, 1, 64);   // value of everything after this lane, applied to 0
    if (lane == 63) tail = 0.f;
    float dnorm = 0.f;
    float sfx = tail;                        // S_j for the lane's last sample
#pragma unroll
    for (int e = SPL - 1; e >= 0; --e) {
        const int j = lane * SPL + e;
        const float wgt = alpha[e] * T[e];
        float4 o = {0.f, 0.f, 0.f, 0.f};
        if (j < c.S) {
            o.x = wgt * gr[0] * col[e][0] * (1.f - col[e][0]);
            o.y = wgt * gr[1] * col[e][1] * (1.f - col[e][1]);
            o.z = wgt * gr[2] * col[e][2] * (1.f - col[e][2]);
            if (j < c.S - 1) {
                const float dalpha = T[e] * (G[e] - sfx);
                const float om = 1.f - alpha[e];            // = exp(-dens*dist)
                o.w = dalpha * dist[e] * om * gate[e];
                const float ddist = dalpha * dens[e] * om;
                dnorm += ddist * (dist[e] / (norm > 0.f ? norm : 1.f));
            }
            *reinterpret_cast<float4*>(draw + ((long long)ray * c.S + j) * 4) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        sfx = G[e] * alpha[e] + (1.f - alpha[e]) * sfx;     // becomes S_{j-1}
    }
    dnorm = wave_sum(dnorm);
    if (drays != nullptr) {     // (the passes of a march run one after the other on one stream)
        if (X.init_drays) {
            if (lane < 11) drays[(long long)ray * 11 + lane] = (lane >= 3 && lane < 6 && norm > 0.f) ? dnorm * c.rays[(long long)ray * 11 + lane] / norm : 0.f;
        } else if (lane < 3 && norm > 0.f) {
            drays[(long long)ray * 11 + 3 + lane] += dnorm * c.rays[(long long)ray * 11 + 3 + lane] / norm;
        }
    }
  }
    if (X.block_max != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        __shared__ float sm[RAYS_PER_BLOCK];
        if (lane == 0) sm[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < RAYS_PER_BLOCK; ++i) m = fmaxf(m, sm[i]);
            X.block_max[blockIdx.x] = m;
        }
    }
}

// {scale, 1/scale} of the loss-scaled fp16 gradient chain from n per-workgroup maxima of |d_raw|: the power of two that puts
// the maximum into [8, 16); 1 when d_raw is all zero or not finite (grad_scale_kernel's rule).  One workgroup.
__global__ __launch_bounds__(256) void loss_scale_kernel(const float* __restrict__ block_max, int n, float* __restrict__ scale2) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = block_max[i];
        m = (v <= 3.402823466e38f) ? fmaxf(m, v) : __int_as_float(0x7f800000);      // (Inf stays Inf; NaN cannot come out of fmaxf)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        float sc = 1.f;
        if (mx > 0.f && mx < 3.0e38f) {
            int e;
            frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
            sc = ldexpf(1.f, 4 - e);        // mx * sc in [8, 16)
        }
        scale2[0] = sc;
        scale2[1] = 1.f / sc;
    }
}

// ------------------------------------------------------ hierarchical sampling
constexpr int SM_MAXN = 512;   // S + Ni rounded up to a power of two

// Bitonic sort of 64 * EPL values held EPL per lane, element e * 64 + lane in v[e], ascending: the same network as the LDS
// form below (so the same result: a sorting network's output is the sorted sequence), but a compare-exchange across lanes is one
// cross-lane read and a min / max instead of two LDS reads, two conditional LDS writes and a wait, partners 64 or more apart are
// registers of the same lane, and everything is unrolled (the LDS form's 28-64 steps ran ~40 instructions each for two elements).
template <int EPL>
__device__ __forceinline__ void wave_bitonic_sort(float (&v)[EPL], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * EPL; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if ((e & (j >> 6)) == 0) {
                        const int pe = e | (j >> 6);
                        const bool up = ((e * 64) & k) == 0;          // (k >= 128 here: the direction depends on e only)
                        const float lo = fminf(v[e], v[pe]), hi = fmaxf(v[e], v[pe]);
                        v[e] = up ? lo : hi;
                        v[pe] = up ? hi : lo;
                    }
                }
            } else {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float o = __shfl_xor(v[e], j, 64);
                    const bool up = ((e * 64 + lane) & k) == 0;
                    v[e] = (lower == up) ? fminf(v[e], o) : fmaxf(v[e], o);
                }
            }
        }
    }
}
__global__ __launch_bounds__(RAYS_PER_BLOCK * 64) void sample_merge_kernel(const float* __restrict__ z,
        const float* __restrict__ weights, int R, int S, int Ni, const float* __restrict__ u,
        float* __restrict__ z_out, float* __restrict__ z_samples, float* __restrict__ z_std, int* __restrict__ flags) {
    __shared__ float s_cdf[RAYS_PER_BLOCK][256];
    __shared__ float s_bins[RAYS_PER_BLOCK][256];
    __shared__ float s_sort[RAYS_PER_BLOCK][SM_MAXN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= R) return;                   // whole wave exits together
    float* cdf = s_cdf[wv];
    float* bins = s_bins[wv];
    float* srt = s_sort[wv];
    const float* zr = z + (long long)ray * S;
    const float* wr = weights + (long long)ray * S;
    const int nb = S - 1;                   // knots in cdf and entries in bins
    // pdf over weights[1..S-2] + 1e-5, inclusive scan in chunks of 64
    float tot = 0.f;
    for (int k = lane; k < S - 2; k += 64) tot += wr[k + 1] + 1e-5f;
    tot = wave_sum(tot);
    float carry = 0.f;
    if (lane == 0) cdf[0] = 0.f;
    for (int k0 = 0; k0 < S - 2; k0 += 64) {
        const int k = k0 + lane;
        const float pdf = k < S - 2 ? (wr[k + 1] + 1e-5f) / tot : 0.f;
        const float inc = wave_incl_sum(pdf, lane) + carry;
        if (k < S - 2) cdf[k + 1] = inc;
        carry = __shfl(inc, 63, 64);
    }
    for (int k = lane; k < nb; k += 64) bins[k] = .5f * (zr[k + 1] + zr[k]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): LDS writes visible to the wave
    // inverse CDF
    float sum = 0.f;
    const int N = S + Ni;
    // whole 64-value rows of coarse depths and of new samples: the union is sorted in registers (wave_bitonic_sort)
    // (a lane keeps four rows of new samples: more than 256 of them take the LDS network below)
    const bool in_regs = (S & 63) == 0 && (Ni & 63) == 0 && Ni <= 256;
    float smp_r[4] = {0.f, 0.f, 0.f, 0.f};            // sample lane + 64 t
    int t_idx = 0;
    for (int i = lane; i < Ni; i += 64, ++t_idx) {
        const float uu = u ? u[(long long)ray * Ni + i] : linspace01(i, Ni);
        int lo = 0, hi = nb;                // first index with cdf > uu  (searchsorted right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (uu - cb) / denom;
        const float bb = bins[below];
        const float smp = bb + t * (bins[above] - bb);
        if (z_samples) z_samples[(long long)ray * Ni + i] = smp;
        srt[S + i] = smp;
        if (t_idx == 0) smp_r[0] = smp; else if (t_idx == 1) smp_r[1] = smp; else if (t_idx == 2) smp_r[2] = smp; else if (t_idx == 3) smp_r[3] = smp;
        sum += smp;
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)Ni;
    float var = 0.f;
    for (int i = lane; i < Ni; i += 64) { const float d = srt[S + i] - mean; var += d * d; }
    var = wave_sum(var);
    if (lane == 0) {
        const float sd = sqrtf(var / (float)Ni);
        if (z_std) z_std[ray] = sd;
        if (flags != nullptr && !(fabsf(sd) <= 3.402823466e38f)) atomicOr(flags, LUSH_FAULT_ZSTD);
    }
    // sort(cat(z, samples)): bitonic network over N2 >= N values padded with +inf
    int N2 = 64;
    while (N2 < N) N2 <<= 1;
    if (in_regs && N2 >= 128) {
        const int sr = S >> 6, nr = Ni >> 6;           // rows of depths, rows of samples
        auto sorted_rows = [&](auto epl_c) {
            constexpr int EPL = decltype(epl_c)::value;
            float v[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float x = __builtin_inff();
                if (e < sr) x = zr[e * 64 + lane];
                else if (e - sr < nr) x = (e - sr) == 0 ? smp_r[0] : (e - sr) == 1 ? smp_r[1] : (e - sr) == 2 ? smp_r[2] : smp_r[3];
                v[e] = x;
            }
            wave_bitonic_sort<EPL>(v, lane);
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (e * 64 < N) z_out[(long long)ray * N + e * 64 + lane] = v[e];      // (N is a multiple of 64 here)
        };
        if (N2 == 128) sorted_rows(std::integral_constant<int, 2>{});
        else if (N2 == 256) sorted_rows(std::integral_constant<int, 4>{});
        else sorted_rows(std::integral_constant<int, 8>{});
        return;
    }
    for (int i = lane; i < S; i += 64) srt[i] = zr[i];
    for (int i = N + lane; i < N2; i += 64) srt[i] = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= N2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < N2; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = srt[i], b = srt[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { srt[i] = b; srt[p] = a; }
                }
            }
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < N; i += 64) z_out[(long long)ray * N + i] = srt[i];
}

// -------------------------------------------------------------- ray prologue
// one ray (o, d) -> its 11 batch columns [o' d' near far viewdir]
__device__ __forceinline__ void pack_one(const float (&o)[3], const float (&d)[3], int ndc, float cx, float cy, float near, float far,
                                         float* __restrict__ b) {
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    b[8] = d[0] / nrm; b[9] = d[1] / nrm; b[10] = d[2] / nrm;
    if (ndc) {   // utils/run_lushnerf_helpers.py:542-562 with near = 1
        const float t = -(1.f + o[2]) / d[2];
        const float px = __fadd_rn(o[0], __fmul_rn(t, d[0])), py = __fadd_rn(o[1], __fmul_rn(t, d[1])),
                    pz = __fadd_rn(o[2], __fmul_rn(t, d[2]));
        b[0] = __fmul_rn(cx, px) / pz;
        b[1] = __fmul_rn(cy, py) / pz;
        b[2] = 1.f + 2.f / pz;
        b[3] = __fmul_rn(cx, __fsub_rn(d[0] / d[2], px / pz));
        b[4] = __fmul_rn(cy, __fsub_rn(d[1] / d[2], py / pz));
        b[5] = -2.f / pz;
    } else {
        b[0] = o[0]; b[1] = o[1]; b[2] = o[2]; b[3] = d[0]; b[4] = d[1]; b[5] = d[2];
    }
    b[6] = near; b[7] = far;
}
// reverse mode of pack_one: g = d loss / d (batch row) -> go, gd (overwritten)
__device__ __forceinline__ void pack_one_bwd(const float (&o)[3], const float (&d)[3], int ndc, float cx, float cy,
                                             const float* __restrict__ g, float (&go)[3], float (&gd)[3]) {
    go[0] = go[1] = go[2] = 0.f;
    gd[0] = gd[1] = gd[2] = 0.f;
    // viewdir = d/|d|
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float v[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};
    const float vg = v[0] * g[8] + v[1] * g[9] + v[2] * g[10];
#pragma unroll
    for (int i = 0; i < 3; ++i) gd[i] += (g[8 + i] - v[i] * vg) / nrm;
    if (ndc) {
        const float t = -(1.f + o[2]) / d[2];
        const float px = o[0] + t * d[0], py = o[1] + t * d[1], pz = o[2] + t * d[2];
        const float ipz = 1.f / pz, ipz2 = ipz * ipz, idz = 1.f / d[2];
        const float gpx = cx * ipz * (g[0] - g[3]);
        const float gpy = cy * ipz * (g[1] - g[4]);
        const float gpz = ipz2 * (-cx * px * (g[0] - g[3]) - cy * py * (g[1] - g[4]) - 2.f * g[2] + 2.f * g[5]);
        gd[0] += cx * g[3] * idz;
        gd[1] += cy * g[4] * idz;
        gd[2] += -(cx * d[0] * g[3] + cy * d[1] * g[4]) * idz * idz;
        const float gp[3] = {gpx, gpy, gpz};
        float gt = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { go[i] += gp[i]; gd[i] += t * gp[i]; gt += gp[i] * d[i]; }
        go[2] += -gt * idz;
        gd[2] += -gt * t * idz;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) { go[i] += g[i]; gd[i] += g[3 + i]; }
    }
}

__global__ void pack_rays_fwd_kernel(const float* __restrict__ rays, int N, int ndc, float cx, float cy, float near,
                                     float far, float* __restrict__ batch) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* r = rays + (long long)n * 6;          // [3][2]: r[2*i] = o_i, r[2*i+1] = d_i
    const float o[3] = {r[0], r[2], r[4]}, d[3] = {r[1], r[3], r[5]};
    pack_one(o, d, ndc, cx, cy, near, far, batch + (long long)n * 11);
}

__global__ void pack_rays_bwd_kernel(const float* __restrict__ rays, int N, int ndc, float cx, float cy,
                                     const float* __restrict__ dbatch, float* __restrict__ drays) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* r = rays + (long long)n * 6;
    const float o[3] = {r[0], r[2], r[4]}, d[3] = {r[1], r[3], r[5]};
    float go[3], gd[3];
    pack_one_bwd(o, d, ndc, cx, cy, dbatch + (long long)n * 11, go, gd);
    float* out = drays + (long long)n * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) { out[2 * i] = go[i]; out[2 * i + 1] = gd[i]; }
}

// ------------------------------------------------------- device-side ray table
// get_rays / get_rays_np (utils/run_lushnerf_helpers.py:517-539) evaluated for N (view, pixel) pairs:
// dirs = [(i + 0.5 - cx)/fx, -(j + 0.5 - cy)/fy, -1], rays_d = R dirs (sum over the last axis of
// dirs[None,:] * c2w[:3,:3]), rays_o = c2w[:3,3].  Replaces the pre-materialised [N_img*H*W, 2, 3] table
// and its per-epoch host permutation (run_lushnerf.py:561-589, 610-614).
__global__ void gen_rays_kernel(const float* __restrict__ c2w, const int64_t* __restrict__ view,
                                const int64_t* __restrict__ px, const int64_t* __restrict__ py, int N, float fx,
                                float fy, float cx, float cy, float* __restrict__ rays) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* M = c2w + view[n] * 12;                       // [3][4] row-major
    const float d0 = ((float)px[n] + (0.5f - cx)) / fx;
    const float d1 = -((float)py[n] + (0.5f - cy)) / fy;
    const float d2 = -1.f;
    float* o = rays + (long long)n * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[2 * i] = M[i * 4 + 3];
        o[2 * i + 1] = (d0 * M[i * 4 + 0] + d1 * M[i * 4 + 1]) + d2 * M[i * 4 + 2];
    }
}

// All H*W pixels of ONE pose in row-major pixel order (the eval path's get_rays, helpers:517-528).
__global__ void gen_rays_image_kernel(const float* __restrict__ c2w, int H, int W, float fx, float fy, float cx,
                                      float cy, float* __restrict__ rays) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= H * W) return;
    const int px = n % W, py = n / W;
    const float d0 = ((float)px + (0.5f - cx)) / fx;
    const float d1 = -((float)py + (0.5f - cy)) / fy;
    const float d2 = -1.f;
    float* o = rays + (long long)n * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[2 * i] = c2w[i * 4 + 3];
        o[2 * i + 1] = (d0 * c2w[i * 4 + 0] + d1 * c2w[i * 4 + 1]) + d2 * c2w[i * 4 + 2];
    }
}

// ------------------------------------------------- consistency branch
// Render_Aligned_Pixel's gather (models/lushnerf.py:958-985): for pose v and sample s the matched
// pixel is (x, y) = align[v][samples[s]][2:4].long(), clamped to the image; its ray is row
// [y][x] of get_rays(H, W, K, c2w[v]).  One thread per (v, s).
__global__ void align_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ align,
                                  const void* __restrict__ cert, int cert_is_u8, const int64_t* __restrict__ samples,
                                  int V, int ns, long long HW, int H, int W, float fx, float fy, float cx, float cy,
                                  float* __restrict__ rays, float* __restrict__ cert_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= V * ns) return;
    const int v = t / ns, s = t % ns;
    long long pix = samples[s];
    pix = pix < 0 ? 0 : (pix >= HW ? HW - 1 : pix);     // memory safety only: the reference would raise an IndexError
    const float* a = align + ((long long)v * HW + pix) * 4;
    long long x = (long long)a[2], y = (long long)a[3];          // .long(): truncation toward zero
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    const float* M = c2w + (long long)v * 12;
    const float d0 = ((float)x + (0.5f - cx)) / fx;
    const float d1 = -((float)y + (0.5f - cy)) / fy;
    const float d2 = -1.f;
    float* o = rays + (long long)t * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[2 * i] = M[i * 4 + 3];
        o[2 * i + 1] = (d0 * M[i * 4 + 0] + d1 * M[i * 4 + 1]) + d2 * M[i * 4 + 2];
    }
    const long long ci = (long long)v * HW + pix;
    cert_out[t] = cert_is_u8 ? (float)(reinterpret_cast<const uint8_t*>(cert)[ci] != 0) : reinterpret_cast<const float*>(cert)[ci];
}

// compute_mean_with_confidence (helpers:665-688) + the masked L1 of run_lushnerf.py:644-650, forward and
// backward (the mean is built with differentiable in-place adds, so gradients flow through it):
//   m[v][p] = cert[v][p] >= thr ; mean[p][c] = sum_v m rgb / max(1, sum_v m)
//   loss = sum_{v,p,c} |rgb - mean| m / #{m}
// One workgroup; thread t owns (p, c) = (t / 3, t % 3).
__global__ __launch_bounds__(256) void consist_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ cert,
                                                          int V, int ns, float thr, float* __restrict__ loss,
                                                          float* __restrict__ grad) {
    __shared__ float s_cnt, s_sum;
    if (threadIdx.x == 0) { s_cnt = 0.f; s_sum = 0.f; }
    __syncthreads();
    float cnt = 0.f;
    for (int i = threadIdx.x; i < V * ns; i += blockDim.x) cnt += cert[i] >= thr ? 1.f : 0.f;
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt != 0.f) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    const float n_mask = s_cnt;                      // 0 -> loss = 0/0 = NaN, as in the reference
    float part = 0.f;
    for (int t = threadIdx.x; t < ns * 3; t += blockDim.x) {
        const int p = t / 3, c = t % 3;
        float sum = 0.f, k = 0.f;
        for (int v = 0; v < V; ++v)
            if (cert[v * ns + p] >= thr) { sum += rgb[(v * ns + p) * 3 + c]; k += 1.f; }
        const float mean = k > 0.f ? sum / k : 0.f;
        float ssum = 0.f;
        for (int v = 0; v < V; ++v)
            if (cert[v * ns + p] >= thr) {
                const float d = rgb[(v * ns + p) * 3 + c] - mean;
                part += fabsf(d);
                ssum += d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            }
        for (int v = 0; v < V; ++v) {
            float g = 0.f;
            if (cert[v * ns + p] >= thr) {
                const float d = rgb[(v * ns + p) * 3 + c] - mean;
                g = ((d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) - ssum / k) / n_mask;
            }
            grad[(v * ns + p) * 3 + c] = g;
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0 && part != 0.f) atomicAdd(&s_sum, part);
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = s_sum / n_mask;
}

// ---------------------------------------------------------------- SE(3) warp
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Se3 { V3 w, nu; float theta, s, c, rn; };   // rn = |r|
__device__ __forceinline__ Se3 se3_setup(V3 r, V3 v) {
    Se3 q;
    q.rn = sqrtf(dot(r, r));
    q.theta = q.rn + 1.0e-10f;
    q.w = (1.f / q.theta) * r;
    q.nu = (1.f / q.theta) * v;
    sincosf(q.theta, &q.s, &q.c);
    return q;
}
// utils/rigid_warping.py:20-45 in closed form: y = R p + t
__device__ __forceinline__ V3 se3_apply(const Se3& q, V3 p) {
    const V3 a = cross(q.w, p), b = cross(q.w, a);
    const V3 e = cross(q.w, q.nu), f = cross(q.w, e);
    return p + q.s * a + (1.f - q.c) * b + q.theta * q.nu + (1.f - q.c) * e + (q.theta - q.s) * f;
}
// reverse mode of se3_apply for one point: adds into gp, and into (gw, gnu, gth, gs, gc)
__device__ __forceinline__ void se3_apply_bwd(const Se3& q, V3 p, V3 gy, V3& gp, V3& gw, V3& gnu, float& gth,
                                              float& gs, float& gc) {
    const V3 a = cross(q.w, p), b = cross(q.w, a);
    const V3 e = cross(q.w, q.nu), f = cross(q.w, e);
    gp = gp + gy;
    gs += dot(gy, a) - dot(gy, f);
    gc += -dot(gy, b) - dot(gy, e);
    gth += dot(gy, q.nu) + dot(gy, f);
    V3 ga = q.s * gy, gb = (1.f - q.c) * gy;
    V3 ge = (1.f - q.c) * gy, gf = (q.theta - q.s) * gy;
    gnu = gnu + q.theta * gy;
    // b = w x a
    gw = gw + cross(a, gb); ga = ga + cross(gb, q.w);
    // a = w x p
    gw = gw + cross(p, ga); gp = gp + cross(ga, q.w);
    // f = w x e
    gw = gw + cross(e, gf); ge = ge + cross(gf, q.w);
    // e = w x nu
    gw = gw + cross(q.nu, ge); gnu = gnu + cross(ge, q.w);
}
// finish: adjoints of (w, nu, theta, s, c) -> (r, v)
__device__ __forceinline__ void se3_finish_bwd(const Se3& q, V3 r, V3 gw, V3 gnu, float gth, float gs, float gc,
                                               V3& gr, V3& gv) {
    gth += gs * q.c - gc * q.s;
    const float it = 1.f / q.theta;
    gth -= it * (dot(gw, q.w) + dot(gnu, q.nu));
    gr = it * gw;
    gv = it * gnu;
    if (q.rn > 0.f) gr = gr + (gth / q.rn) * r;
}

// acts layout per image (LUSH_RBK_ACT_STRIDE floats)
constexpr int RA_E = 0, RA_H0 = 64, RA_HR = 320, RA_HV = 352, RA_HW = 384, RA_WS = 416, RA_R = 424, RA_V = 440,
              RA_WN = 456;

__global__ void rbk_warp_fwd_kernel(const float* __restrict__ rays, const int64_t* __restrict__ idx, int N, int M,
                                    const float* __restrict__ acts, float* __restrict__ new_rays,
                                    float* __restrict__ ccw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* r6 = rays + (long long)n * 6;
    const V3 o = {r6[0], r6[2], r6[4]}, d = {r6[1], r6[3], r6[5]};
    const V3 end = o + d;
    const float* A = acts + idx[n] * LUSH_RBK_ACT_STRIDE;
    float* out = new_rays + (long long)n * (M + 1) * 6;
    out[0] = o.x; out[1] = d.x; out[2] = o.y; out[3] = d.y; out[4] = o.z; out[5] = d.z;
    for (int m = 0; m < M; ++m) {   // r.reshape(N,3,M): component c of motion m is column c*M+m (models/lushnerf.py:76)
        const V3 r = {A[RA_R + m], A[RA_R + M + m], A[RA_R + 2 * M + m]};
        const V3 v = {A[RA_V + m], A[RA_V + M + m], A[RA_V + 2 * M + m]};
        const Se3 q = se3_setup(r, v);
        const V3 wo = se3_apply(q, o), we = se3_apply(q, end);
        const V3 wd = we - wo;
        float* w6 = out + (m + 1) * 6;
        w6[0] = wo.x; w6[1] = wd.x; w6[2] = wo.y; w6[3] = wd.y; w6[4] = wo.z; w6[5] = wd.z;
    }
    for (int m = 0; m <= M; ++m) ccw[(long long)n * (M + 1) + m] = A[RA_WN + m];
}

__global__ void rbk_warp_bwd_kernel(const float* __restrict__ rays, const int64_t* __restrict__ idx, int N, int M,
                                    const float* __restrict__ acts, const float* __restrict__ dnew,
                                    const float* __restrict__ dccw, const uint8_t* __restrict__ mask,
                                    float* __restrict__ d_rvw, int rvw_stride, float* __restrict__ drays) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const long long img = idx[n];
    float* G = d_rvw + img * rvw_stride;
    if (dccw)
        for (int m = 0; m <= M; ++m) atomicAdd(G + 24 + m, dccw[(long long)n * (M + 1) + m]);
    const bool live = dnew != nullptr && (mask == nullptr || mask[n] != 0);
    V3 go = {0.f, 0.f, 0.f}, gd = {0.f, 0.f, 0.f};
    if (live) {
        const float* r6 = rays + (long long)n * 6;
        const V3 o = {r6[0], r6[2], r6[4]}, d = {r6[1], r6[3], r6[5]};
        const V3 end = o + d;
        const float* A = acts + img * LUSH_RBK_ACT_STRIDE;
        const float* g6 = dnew + (long long)n * (M + 1) * 6;
        go = {g6[0], g6[2], g6[4]};
        gd = {g6[1], g6[3], g6[5]};
        for (int m = 0; m < M; ++m) {
            const V3 r = {A[RA_R + m], A[RA_R + M + m], A[RA_R + 2 * M + m]};
            const V3 v = {A[RA_V + m], A[RA_V + M + m], A[RA_V + 2 * M + m]};
            const Se3 q = se3_setup(r, v);
            const float* w6 = g6 + (m + 1) * 6;
            const V3 gwo = {w6[0], w6[2], w6[4]}, gwd = {w6[1], w6[3], w6[5]};
            // wd = we - wo
            const V3 gye = gwd, gyo = gwo - gwd;
            V3 gp_o = {0, 0, 0}, gp_e = {0, 0, 0}, gw = {0, 0, 0}, gnu = {0, 0, 0};
            float gth = 0.f, gs = 0.f, gc = 0.f;
            se3_apply_bwd(q, o, gyo, gp_o, gw, gnu, gth, gs, gc);
            se3_apply_bwd(q, end, gye, gp_e, gw, gnu, gth, gs, gc);
            V3 gr, gv;
            se3_finish_bwd(q, r, gw, gnu, gth, gs, gc, gr, gv);
            go = go + gp_o + gp_e;
            gd = gd + gp_e;
            atomicAdd(G + m, gr.x); atomicAdd(G + M + m, gr.y); atomicAdd(G + 2 * M + m, gr.z);
            atomicAdd(G + 12 + m, gv.x); atomicAdd(G + 12 + M + m, gv.y); atomicAdd(G + 12 + 2 * M + m, gv.z);
        }
    }
    if (drays) {
        float* out = drays + (long long)n * 6;
        out[0] = go.x; out[1] = gd.x; out[2] = go.y; out[3] = gd.y; out[4] = go.z; out[5] = gd.z;
    }
}

// ------------------------------------------------- warp + NDC + pack in one (SURVEY 7.2 `rbk_warp_ndc`)
// Rigid_Blurring_Kernel.rbk_warp (models/lushnerf.py:75-98) followed by the head of render_train_scene (:772-795: view
// directions, ndc_rays helpers:542-562, near / far columns) for the M + 1 rays of every input ray, and the same head for the
// input ray alone (render_train_noise, :827-850): rays [N][3][2] -> batch [N*(M+1)][11], ccw [N][M+1], batch0 [N][11].
// One thread per (input ray, motion slot): the warped rays never exist in memory.
__global__ void rbk_warp_ndc_fwd_kernel(const float* __restrict__ rays, const int64_t* __restrict__ idx, int N, int M,
                                        const float* __restrict__ acts, int ndc, float cx, float cy, float near, float far,
                                        float* __restrict__ batch, float* __restrict__ ccw, float* __restrict__ batch0) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M1 = M + 1;
    if (t >= (long long)N * M1) return;
    const long long n = t / M1;
    const int m = (int)(t % M1);
    const float* r6 = rays + n * 6;
    const V3 o = {r6[0], r6[2], r6[4]}, d = {r6[1], r6[3], r6[5]};
    const float* A = acts + idx[n] * LUSH_RBK_ACT_STRIDE;
    V3 wo = o, wd = d;
    if (m > 0) {    // r.reshape(N,3,M): component c of motion m-1 is column c*M + m-1 (models/lushnerf.py:76)
        const int k = m - 1;
        const V3 r = {A[RA_R + k], A[RA_R + M + k], A[RA_R + 2 * M + k]};
        const V3 v = {A[RA_V + k], A[RA_V + M + k], A[RA_V + 2 * M + k]};
        const Se3 q = se3_setup(r, v);
        wo = se3_apply(q, o);
        wd = se3_apply(q, o + d) - wo;
    }
    const float oo[3] = {wo.x, wo.y, wo.z}, dd[3] = {wd.x, wd.y, wd.z};
    pack_one(oo, dd, ndc, cx, cy, near, far, batch + t * 11);
    ccw[t] = A[RA_WN + m];
    if (m == 0 && batch0 != nullptr) pack_one(oo, dd, ndc, cx, cy, near, far, batch0 + n * 11);
}

// Reverse: dbatch [N*(M+1)][11] (may be NULL: only the weights' gradient), dccw [N][M+1] (may be NULL) -> d_rvw [num_img][stride]
// (accumulated; zeroed by lush_rbk_mlp_fwd), drays [N][3][2] (overwritten; may be NULL).  mask as in rbk_warp_bwd_kernel.
// One thread per (input ray, motion slot), RPB rays per workgroup.  The per-image sums are formed in LDS first (up to WN_IMGS
// images) and leave as one global atomic per touched (image, component) and workgroup: with 4096 rays of 26 images the
// first version's 119 k atomics on 754 words took 33 us.
constexpr int WN_IMGS = 64;
__global__ __launch_bounds__(1024) void rbk_warp_ndc_bwd_kernel(const float* __restrict__ rays, const int64_t* __restrict__ idx, int N, int M,
                                        const float* __restrict__ acts, int ndc, float cx, float cy,
                                        const float* __restrict__ dbatch, const float* __restrict__ dccw,
                                        const uint8_t* __restrict__ mask, float* __restrict__ d_rvw, int rvw_stride,
                                        float* __restrict__ drays, int num_img, int RPB) {
    __shared__ float tab[WN_IMGS][LUSH_RBK_RVW_STRIDE];
    __shared__ float racc[1024 / 2][6];           // d(o, d) of the workgroup's rays (RPB <= 512: M >= 1)
    const int M1 = M + 1;
    const bool in_lds = num_img <= WN_IMGS;
    for (int i = threadIdx.x; i < WN_IMGS * LUSH_RBK_RVW_STRIDE; i += blockDim.x) (&tab[0][0])[i] = 0.f;
    for (int i = threadIdx.x; i < RPB * 6; i += blockDim.x) (&racc[0][0])[i] = 0.f;
    __syncthreads();
    const int lr = threadIdx.x / M1, slot = threadIdx.x % M1;
    const long long n = (long long)blockIdx.x * RPB + lr;
    const long long img = (lr < RPB && n < N) ? idx[n] : -1;
    if (img >= 0 && img < num_img) {          // (an index outside the table contributes nothing instead of writing outside it)
        float* G = in_lds ? &tab[img][0] : d_rvw + img * rvw_stride;
        if (dccw) atomicAdd(G + 24 + slot, dccw[n * M1 + slot]);
        const bool live = dbatch != nullptr && (mask == nullptr || mask[n] != 0);
        if (live) {
            const float* r6 = rays + n * 6;
            const V3 o = {r6[0], r6[2], r6[4]}, d = {r6[1], r6[3], r6[5]};
            const float* g11 = dbatch + (n * M1 + slot) * 11;
            V3 go, gd;
            if (slot == 0) {    // the input ray itself
                const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
                float a[3], b[3];
                pack_one_bwd(oo, dd, ndc, cx, cy, g11, a, b);
                go = {a[0], a[1], a[2]};
                gd = {b[0], b[1], b[2]};
            } else {
                const int m = slot - 1;
                const V3 end = o + d;
                const float* A = acts + img * LUSH_RBK_ACT_STRIDE;
                const V3 r = {A[RA_R + m], A[RA_R + M + m], A[RA_R + 2 * M + m]};
                const V3 v = {A[RA_V + m], A[RA_V + M + m], A[RA_V + 2 * M + m]};
                const Se3 q = se3_setup(r, v);
                const V3 wo = se3_apply(q, o), wd = se3_apply(q, end) - wo;     // (recomputed: the warped rays were never stored)
                const float oo[3] = {wo.x, wo.y, wo.z}, dd[3] = {wd.x, wd.y, wd.z};
                float a[3], b[3];
                pack_one_bwd(oo, dd, ndc, cx, cy, g11, a, b);
                const V3 gwo = {a[0], a[1], a[2]}, gwd = {b[0], b[1], b[2]};
                const V3 gye = gwd, gyo = gwo - gwd;       // wd = we - wo
                V3 gp_o = {0, 0, 0}, gp_e = {0, 0, 0}, gw = {0, 0, 0}, gnu = {0, 0, 0};
                float gth = 0.f, gs = 0.f, gc = 0.f;
                se3_apply_bwd(q, o, gyo, gp_o, gw, gnu, gth, gs, gc);
                se3_apply_bwd(q, end, gye, gp_e, gw, gnu, gth, gs, gc);
                V3 gr, gv;
                se3_finish_bwd(q, r, gw, gnu, gth, gs, gc, gr, gv);
                go = gp_o + gp_e;
                gd = gp_e;
                atomicAdd(G + m, gr.x); atomicAdd(G + M + m, gr.y); atomicAdd(G + 2 * M + m, gr.z);
                atomicAdd(G + 12 + m, gv.x); atomicAdd(G + 12 + M + m, gv.y); atomicAdd(G + 12 + 2 * M + m, gv.z);
            }
            if (drays) {
                atomicAdd(&racc[lr][0], go.x); atomicAdd(&racc[lr][1], gd.x); atomicAdd(&racc[lr][2], go.y);
                atomicAdd(&racc[lr][3], gd.y); atomicAdd(&racc[lr][4], go.z); atomicAdd(&racc[lr][5], gd.z);
            }
        }
    }
    __syncthreads();
    if (in_lds)
        for (int i = threadIdx.x; i < num_img * LUSH_RBK_RVW_STRIDE; i += blockDim.x) {
            const float v = (&tab[0][0])[i];
            if (v != 0.f) atomicAdd(d_rvw + (long long)(i / LUSH_RBK_RVW_STRIDE) * rvw_stride + i % LUSH_RBK_RVW_STRIDE, v);
        }
    if (drays)
        for (int i = threadIdx.x; i < RPB * 6; i += blockDim.x) {
            const long long nn = (long long)blockIdx.x * RPB + i / 6;
            if (nn < N) drays[nn * 6 + i % 6] = (&racc[0][0])[i];
        }
}

// ------------------------------------------------------------------- RBK MLP
// y[i][o] = act(b[o] + sum_k W[o][k] x[i][k]) for a few dozen images through ten small dense stages (4 MFLOP in all).
// A workgroup of RBK_NT threads takes RBK_IPB images; their activation rows (512 floats each) live in LDS for the whole
// kernel with an ODD row stride (RBK_LS), and every stage's matrix is copied into LDS once at the head of the kernel.
//
// What these kernels cost was never arithmetic or LDS bandwidth but DEPENDENT LATENCIES IN SERIES, one wave per SIMD with
// nothing to switch to: as `for (t = threadIdx.x; t < count; t += blockDim.x) lds[..] = W[t]` with a run-time trip count the
// compiler emits load, s_waitcnt vmcnt(0), ds_write per iteration, so the 23.8 K weight floats were 93 global-memory round
// trips one after the other (0.3-0.4 us each: the 37 us of the forward, whatever the number of workgroups, staging depth or
// summation batching -- round 4 measured five such variants at 36-39 us before finding this), and the per-image sums of
// the backward's weight gradients were 16 x 4 LDS round trips in series per stage.  Hence the compile-time thread count and
// image count below: every loop has a constant trip count, is fully unrolled with its loads in front of its uses, and rows
// of absent images (n < RBK_IPB) hold zeros instead of shortening a loop.
constexpr int RBK_LS = LUSH_RBK_ACT_STRIDE + 1;
constexpr int RBK_NT = 256;       // threads per workgroup
constexpr int RBK_IPB = 4;        // images per workgroup
constexpr int RBK_WALL = 4 * 64 * 65 + 3 * 32 * 65 + 29 * 33;      // floats of every stage's matrix at row pitch IN + 1 (num_motion <= 4)
constexpr int RBK_FETCH = 16;     // floats per thread of one matrix fetch (64 x 64 / RBK_NT)

// global -> registers (up to RBK_FETCH x RBK_NT floats of W, coalesced), then registers -> LDS at row pitch IN + 1: two calls,
// so that the fetches of SEVERAL matrices are all in flight before the first one is waited for
template <int IN>
__device__ __forceinline__ void rbk_fetch(const float* __restrict__ W, int OUT, float (&v)[RBK_FETCH]) {
#pragma unroll
    for (int j = 0; j < RBK_FETCH; ++j) {
        const int t = j * RBK_NT + threadIdx.x;
        v[j] = t < OUT * IN ? W[t] : 0.f;
    }
}
template <int IN>
__device__ __forceinline__ void rbk_put(float* __restrict__ wb, int OUT, const float (&v)[RBK_FETCH]) {
#pragma unroll
    for (int j = 0; j < RBK_FETCH; ++j) {
        const int t = j * RBK_NT + threadIdx.x;
        if (t < OUT * IN) wb[(t / IN) * (IN + 1) + t % IN] = v[j];
    }
}
struct RbkStages {       // the ten matrices of one pass in the order the pass uses them
    const float* W[10];
    const float* B[10];      // biases (forward only): kept in the pad column of the staged matrix, wb[o * (IN + 1) + IN]
    int OUT[10];
    int WO[10];              // float offset of the staged matrix in the wall
};
// IN of stage st is 64 for st in [first64, first64 + 7), 32 for the other three (forward: 64s first; backward: 32s first)
template <bool BIAS, int FIRST64>
__device__ __forceinline__ void rbk_stage_all(RbkStages& S, float* __restrict__ wall) {
    int off = 0;
#pragma unroll
    for (int st = 0; st < 10; ++st) {
        S.WO[st] = off;
        off += S.OUT[st] * ((st >= FIRST64 && st < FIRST64 + 7 ? 64 : 32) + 1);
    }
    float bias[10];
    if (BIAS) {
#pragma unroll
        for (int st = 0; st < 10; ++st) bias[st] = (int)threadIdx.x < S.OUT[st] ? S.B[st][threadIdx.x] : 0.f;
    }
    // every matrix's fetch in flight before the first is waited for (160 registers of fetched weights; one wave per SIMD has 512)
    {
        float v[10][RBK_FETCH];
#pragma unroll
        for (int st = 0; st < 10; ++st) {
            if (st >= FIRST64 && st < FIRST64 + 7) rbk_fetch<64>(S.W[st], S.OUT[st], v[st]);
            else rbk_fetch<32>(S.W[st], S.OUT[st], v[st]);
        }
#pragma unroll
        for (int st = 0; st < 10; ++st) {
            if (st >= FIRST64 && st < FIRST64 + 7) rbk_put<64>(wall + S.WO[st], S.OUT[st], v[st]);
            else rbk_put<32>(wall + S.WO[st], S.OUT[st], v[st]);
        }
    }
    if (BIAS) {
#pragma unroll
        for (int st = 0; st < 10; ++st) {
            const int in = st >= FIRST64 && st < FIRST64 + 7 ? 64 : 32;
            if ((int)threadIdx.x < S.OUT[st]) wall[S.WO[st] + threadIdx.x * (in + 1) + in] = bias[st];
        }
    }
}

// y[i][o] = act(b[o] + sum_k W[o][k] x[i][k]) from the staged W: bias first, k ascending (the order of every earlier version);
// RBK_IPB x OUT <= RBK_NT outputs, one per thread, image index fastest across lanes
template <int IN>
__device__ __forceinline__ void rbk_dense_lds(const float* __restrict__ wb, const float* x, float* y, int OUT, int relu) {
    const int i = threadIdx.x % RBK_IPB, o = threadIdx.x / RBK_IPB;
    if (o >= OUT) return;
    const float* w = wb + o * (IN + 1);
    const float* xi = x + i * RBK_LS;
    float s = w[IN];
#pragma unroll
    for (int k0 = 0; k0 < IN; k0 += 16) {       // (operands of 16 terms fetched before they are summed)
        float wv[16], xv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { wv[j] = w[k0 + j]; xv[j] = xi[k0 + j]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) s += wv[j] * xv[j];
    }
    y[i * RBK_LS + o] = relu ? fmaxf(s, 0.f) : s;
}

__global__ __launch_bounds__(RBK_NT) void rbk_mlp_fwd_kernel(lush_rbk_params p, int n_all, int M, float window,
                                                            float* __restrict__ acts) {
    const int img0 = blockIdx.x * RBK_IPB;
    const int n = n_all - img0 < RBK_IPB ? n_all - img0 : RBK_IPB;
    p.embed += (long long)img0 * 64;
    acts += (long long)img0 * LUSH_RBK_ACT_STRIDE;
    extern __shared__ float rbk_lds[];
    float* A = rbk_lds;                             // [RBK_IPB][RBK_LS] activations
    float* wall = rbk_lds + RBK_IPB * RBK_LS;       // every stage's matrix
    constexpr int ST = LUSH_RBK_ACT_STRIDE;
    RbkStages S = {{p.w_trunk[0], p.w_trunk[1], p.w_trunk[2], p.w_trunk[3], p.w_rb, p.w_vb, p.w_wb, p.w_r, p.w_v, p.w_w},
                   {p.b_trunk[0], p.b_trunk[1], p.b_trunk[2], p.b_trunk[3], p.b_rb, p.b_vb, p.b_wb, p.b_r, p.b_v, p.b_w},
                   {64, 64, 64, 64, 32, 32, 32, 3 * M, 3 * M, M + 1}, {}};
    const int XO[10] = {RA_E, RA_H0, RA_H0 + 64, RA_H0 + 128, RA_H0 + 192, RA_H0 + 192, RA_H0 + 192, RA_HR, RA_HV, RA_HW};
    const int YO[10] = {RA_H0, RA_H0 + 64, RA_H0 + 128, RA_H0 + 192, RA_HR, RA_HV, RA_HW, RA_R, RA_V, RA_WS};
    {   // (RBK_IPB x 64 = RBK_NT: one embedding float per thread; an absent image's row is zeros and is never written out)
        const int i = threadIdx.x / 64;
        A[i * RBK_LS + RA_E + threadIdx.x % 64] = i < n ? p.embed[threadIdx.x] : 0.f;
    }
    rbk_stage_all<true, 0>(S, wall);
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 10; ++st) {
        if (st < 7) rbk_dense_lds<64>(wall + S.WO[st], A + XO[st], A + YO[st], S.OUT[st], 1);
        else rbk_dense_lds<32>(wall + S.WO[st], A + XO[st], A + YO[st], S.OUT[st], 0);
        // (the three branches and the three heads read one input and write disjoint outputs: one barrier per group)
        if (st != 4 && st != 5 && st != 7 && st != 8) __syncthreads();
    }
    if ((int)threadIdx.x < RBK_IPB * 3 * M) {
        const int i = threadIdx.x / (3 * M), o = threadIdx.x % (3 * M);
        A[i * RBK_LS + RA_R + o] *= window;
        A[i * RBK_LS + RA_V + o] *= window;
    }
    if ((int)threadIdx.x < RBK_IPB) {
        const int i = threadIdx.x;
        float sum = 0.f;
        for (int m = 0; m <= M; ++m) {
            const float s = 1.f / (1.f + expf(-A[i * RBK_LS + RA_WS + m]));
            A[i * RBK_LS + RA_WS + m] = s;
            sum += s;
        }
        for (int m = 0; m <= M; ++m) A[i * RBK_LS + RA_WN + m] = A[i * RBK_LS + RA_WS + m] / (sum + 1e-10f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RBK_IPB * ST / RBK_NT; ++j) {
        const int t = j * RBK_NT + threadIdx.x;
        if (t < n * ST) acts[t] = (t % ST) < RA_WN + 8 ? A[(t / ST) * RBK_LS + (t % ST)] : 0.f;
    }
}

// dx[i][k] = [gate[i][k] > 0] * (sum_o W[o][k] dz[i][o] (+ dx[i][k])) from the staged W (row pitch IN + 1); o ascending, the
// previous value added last, the gate applied to the total -- the arithmetic of round 3's separate passes.
// RBK_IPB x IN <= RBK_NT outputs, one per thread.
template <int IN>
__device__ __forceinline__ void rbk_dense_bwd_x_lds(const float* __restrict__ wb, const float* dz, float* dx, const float* gate,
                                                    int OUT, int accumulate) {
    const int i = threadIdx.x % RBK_IPB, k = threadIdx.x / RBK_IPB;
    if (k >= IN) return;
    float s = 0.f;
    const float* zi = dz + i * RBK_LS;
    int o0 = 0;
    for (; o0 + 8 <= OUT; o0 += 8) {          // (operands of 8 terms fetched before they are summed; same order of the sum)
        float wv[8], zv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[j] = wb[(o0 + j) * (IN + 1) + k]; zv[j] = zi[o0 + j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += wv[j] * zv[j];
    }
    if (o0 < OUT) {                            // the heads' 3 M or M + 1 outputs: up to 7 more, fetched together as well
        float wv[8], zv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = o0 + j < OUT;
            wv[j] = in ? wb[(o0 + j) * (IN + 1) + k] : 0.f;
            zv[j] = in ? zi[o0 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (o0 + j < OUT) s += wv[j] * zv[j];
    }
    if (accumulate) s += dx[i * RBK_LS + k];
    if (gate && !(gate[i * RBK_LS + k] > 0.f)) s = 0.f;
    dx[i * RBK_LS + k] = s;
}
// dW[o][k] += sum_i dz[i][o] x[i][k], db[o] += sum_i dz[i][o]: added into buffers that hold values (the trainer's flat gradient,
// or zeros the launcher wrote), by atomics because several workgroups -- each with its own images -- add into the same matrix.
// Thread t takes outputs t, t + RBK_NT, ...: k = t % IN is the same for all of them (RBK_NT is a multiple of IN), so a thread
// reads its RBK_IPB activations once; rows of absent images hold zeros.
template <int IN>
__device__ __forceinline__ void rbk_dense_bwd_w(const float* dz, const float* x, float* __restrict__ dW, float* __restrict__ db, int OUT) {
    const int k = threadIdx.x % IN, o_first = threadIdx.x / IN;
    constexpr int OSTEP = RBK_NT / IN;
    float xv[RBK_IPB];
#pragma unroll
    for (int i = 0; i < RBK_IPB; ++i) xv[i] = x[i * RBK_LS + k];
#pragma unroll
    for (int r = 0; r < 64 / OSTEP; ++r) {
        const int o = o_first + r * OSTEP;
        if (o < OUT) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < RBK_IPB; ++i) s += dz[i * RBK_LS + o] * xv[i];
            atomicAdd(dW + o * IN + k, s);
        }
    }
    if ((int)threadIdx.x < OUT) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RBK_IPB; ++i) s += dz[i * RBK_LS + threadIdx.x];
        atomicAdd(db + threadIdx.x, s);
    }
    // (no barrier: dW / db are outputs only; the adjoints a later stage overwrites are guarded by that stage's barrier)
}

// Workgroup b takes images [b RBK_IPB, (b + 1) RBK_IPB): d(activation) is per image, the weight gradients are sums over images.
__global__ __launch_bounds__(RBK_NT) void rbk_mlp_bwd_kernel(lush_rbk_params p, int n_all, int M, float window,
                                                            const float* __restrict__ acts,
                                                            float* __restrict__ d_rvw, lush_rbk_grads g,
                                                            int RS /* floats between two images' rows of d_rvw */) {
    const int img0 = blockIdx.x * RBK_IPB;
    const int n = n_all - img0 < RBK_IPB ? n_all - img0 : RBK_IPB;
    acts += (long long)img0 * LUSH_RBK_ACT_STRIDE;
    d_rvw += (long long)img0 * RS;
    g.embed += (long long)img0 * 64;
    // LDS: activations [RBK_IPB][RBK_LS], then the adjoints of the pre-activations in the same per-image layout, then the matrices
    extern __shared__ float rbk_lds[];
    float* A = rbk_lds;
    float* sc = rbk_lds + RBK_IPB * RBK_LS;
    float* wall = rbk_lds + 2 * RBK_IPB * RBK_LS;
    constexpr int ST = LUSH_RBK_ACT_STRIDE;
    {
        float v[RBK_IPB * ST / RBK_NT];
#pragma unroll
        for (int j = 0; j < RBK_IPB * ST / RBK_NT; ++j) {
            const int t = j * RBK_NT + threadIdx.x;
            v[j] = t < n * ST ? acts[t] : 0.f;            // (an absent image: zero activations, zero adjoints)
        }
#pragma unroll
        for (int j = 0; j < RBK_IPB * ST / RBK_NT; ++j) {
            const int t = j * RBK_NT + threadIdx.x;
            A[(t / ST) * RBK_LS + (t % ST)] = v[j];
            sc[(t / ST) * RBK_LS + (t % ST)] = 0.f;
        }
    }
    // ten dx stages: heads r, v, w; branches rb, vb, wb into one d h3; trunk 3..0
    RbkStages S = {{p.w_r, p.w_v, p.w_w, p.w_rb, p.w_vb, p.w_wb, p.w_trunk[3], p.w_trunk[2], p.w_trunk[1], p.w_trunk[0]}, {},
                   {3 * M, 3 * M, M + 1, 32, 32, 32, 64, 64, 64, 64}, {}};
    rbk_stage_all<false, 3>(S, wall);
    __syncthreads();
    if ((int)threadIdx.x < RBK_IPB * 3 * M) {
        const int i = threadIdx.x / (3 * M), o = threadIdx.x % (3 * M);
        if (i < n) {
            sc[i * RBK_LS + RA_R + o] = d_rvw[i * RS + o] * window;
            sc[i * RBK_LS + RA_V + o] = d_rvw[i * RS + 12 + o] * window;
            // consumed: every element of d_rvw is read by exactly one thread, which leaves it ZERO -- the rows live in the zero
            // tail of `acts` (LUSH_RBK_RVW_OFFSET), so a second backward over the same saved activations starts from zero again
            d_rvw[i * RS + o] = 0.f;
            d_rvw[i * RS + 12 + o] = 0.f;
        }
    }
    if ((int)threadIdx.x < n) {
        const int i = threadIdx.x;
        float sum = 1e-10f, dotp = 0.f;
        for (int m = 0; m <= M; ++m) { sum += A[i * RBK_LS + RA_WS + m]; dotp += d_rvw[i * RS + 24 + m] * A[i * RBK_LS + RA_WS + m]; }
        for (int m = 0; m <= M; ++m) {
            const float ws = A[i * RBK_LS + RA_WS + m];
            const float dws = d_rvw[i * RS + 24 + m] / sum - dotp / (sum * sum);
            sc[i * RBK_LS + RA_WS + m] = dws * ws * (1.f - ws);
            d_rvw[i * RS + 24 + m] = 0.f;
        }
    }
    __syncthreads();
    const int ZO[10] = {RA_R, RA_V, RA_WS, RA_HR, RA_HV, RA_HW, RA_H0 + 192, RA_H0 + 128, RA_H0 + 64, RA_H0};
    const int XO[10] = {RA_HR, RA_HV, RA_HW, RA_H0 + 192, RA_H0 + 192, RA_H0 + 192, RA_H0 + 128, RA_H0 + 64, RA_H0, RA_E};
    const int GATE[10] = {1, 1, 1, 0, 0, 1, 1, 1, 1, 0};       // ReLU gate of the stage's input activations (d h3: after the third addend)
    const int ACC[10] = {0, 0, 0, 0, 1, 1, 0, 0, 0, 0};
    float* gw[10] = {g.w_r, g.w_v, g.w_w, g.w_rb, g.w_vb, g.w_wb, g.w_trunk[3], g.w_trunk[2], g.w_trunk[1], g.w_trunk[0]};
    float* gb[10] = {g.b_r, g.b_v, g.b_w, g.b_rb, g.b_vb, g.b_wb, g.b_trunk[3], g.b_trunk[2], g.b_trunk[1], g.b_trunk[0]};
#pragma unroll
    for (int st = 0; st < 10; ++st) {
        // dW / db of this stage: dz (complete since the previous barrier) x the stage's input activations
        if (st < 3) {
            rbk_dense_bwd_w<32>(sc + ZO[st], A + XO[st], gw[st], gb[st], S.OUT[st]);
            rbk_dense_bwd_x_lds<32>(wall + S.WO[st], sc + ZO[st], sc + XO[st], GATE[st] ? A + XO[st] : nullptr, S.OUT[st], ACC[st]);
        } else {
            rbk_dense_bwd_w<64>(sc + ZO[st], A + XO[st], gw[st], gb[st], S.OUT[st]);
            rbk_dense_bwd_x_lds<64>(wall + S.WO[st], sc + ZO[st], sc + XO[st], GATE[st] ? A + XO[st] : nullptr, S.OUT[st], ACC[st]);
        }
        // (the three heads write three different adjoints: one barrier behind them; the three branches accumulate into ONE d h3
        // in order, a barrier each)
        if (st != 0 && st != 1) __syncthreads();
    }
    {   // (an image's embedding row belongs to one workgroup; RBK_IPB x 64 = RBK_NT)
        const int i = threadIdx.x / 64;
        if (i < n) g.embed[threadIdx.x] += sc[i * RBK_LS + RA_E + (threadIdx.x % 64)];
    }
}

// ------------------------------------------------------- blur mix / tone map
__global__ void wsum_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ccw, int N, int M, int C,
                                float* __restrict__ y) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * C) return;
    const long long n = t / C; const int c = (int)(t % C);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += x[(n * M + m) * C + c] * ccw[n * M + m];
    y[t] = s;
}
__global__ void wsum_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ccw, int N, int M, int C,
                                const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dccw) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (n, m)
    if (t >= (long long)N * M) return;
    const long long n = t / M;
    float s = 0.f;
    const float w = ccw[t];
    for (int c = 0; c < C; ++c) {
        const float g = dy[n * C + c];
        s += g * x[t * C + c];
        if (dx) dx[t * C + c] = g * w;
    }
    if (dccw) dccw[t] = s;
}

constexpr float INV_GAMMA = (float)(1.0 / 2.2);
__global__ void tonemap_fwd_kernel(const float* __restrict__ x, const float* __restrict__ nraw, int n3, int gamma,
                                   float* __restrict__ y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    float v = x[t];
    if (nraw) v += 0.1f * (1.f / (1.f + expf(-nraw[t])));
    y[t] = gamma ? powf(v, INV_GAMMA) : v;
}
__global__ void tonemap_bwd_kernel(const float* __restrict__ x, const float* __restrict__ nraw, int n3, int gamma,
                                   const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dnraw) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n3) return;
    float v = x[t], sg = 0.f;
    if (nraw) { sg = 1.f / (1.f + expf(-nraw[t])); v += 0.1f * sg; }
    const float gv = gamma ? dy[t] * INV_GAMMA * powf(v, INV_GAMMA - 1.f) : dy[t];
    if (dx) dx[t] = gv;
    if (nraw && dnraw) dnraw[t] = gv * 0.1f * sg * (1.f - sg);
}
__global__ void noise_act_fwd_kernel(const float* __restrict__ x, int n, float* __restrict__ y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = 0.1f * (1.f / (1.f + expf(-x[t])));
}
__global__ void noise_act_bwd_kernel(const float* __restrict__ x, int n, const float* __restrict__ dy,
                                     float* __restrict__ dx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { const float s = 1.f / (1.f + expf(-x[t])); dx[t] = dy[t] * 0.1f * s * (1.f - s); }
}
// ------------------------------------------------- blur mix + noise + tone map in one (SURVEY 7.2 `blur_mix_tonemap`)
// The tail of NeRFAll.forward's training branch (models/lushnerf.py:644-654): rbk_weighted_sum of the fine and the coarse
// colour (:100-116), rgb_noise = 0.1 sigmoid(noise_raw) (:649), and the five tone-mapped / plain outputs of the 7-tuple
//   blur = tm(sum_m ccw x + rgb_noise), blur0 likewise, noise = rgb_noise, sharp = tm(sum_m ccw x), sharp0 likewise.
// One thread per (input ray, channel); the weighted sums in the order of wsum_fwd_kernel (m ascending).
__device__ __forceinline__ float tm_apply(float v, int gamma) { return gamma ? powf(v, INV_GAMMA) : v; }
__device__ __forceinline__ float tm_slope(float v, int gamma) { return gamma ? INV_GAMMA * powf(v, INV_GAMMA - 1.f) : 1.f; }
__global__ void blur_mix_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb0, const float* __restrict__ ccw,
                                    const float* __restrict__ nraw, int nraw_ld, int N, int M1, int gamma, float* __restrict__ blur,
                                    float* __restrict__ blur0, float* __restrict__ noise, float* __restrict__ sharp,
                                    float* __restrict__ sharp0) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * 3) return;
    const long long n = t / 3;
    const int c = (int)(t % 3);
    float s = 0.f, s0 = 0.f;
    for (int m = 0; m < M1; ++m) {
        const float w = ccw[n * M1 + m];
        s += rgb[(n * M1 + m) * 3 + c] * w;
        s0 += rgb0[(n * M1 + m) * 3 + c] * w;
    }
    const float nz = 0.1f * (1.f / (1.f + expf(-nraw[n * nraw_ld + c])));
    blur[t] = tm_apply(s + nz, gamma);
    blur0[t] = tm_apply(s0 + nz, gamma);
    noise[t] = nz;
    sharp[t] = tm_apply(s, gamma);
    sharp0[t] = tm_apply(s0, gamma);
}
// Reverse: one thread per (input ray, motion slot).  Any of the five output gradients may be NULL (= 0).  d_rgb / d_rgb0
// [N*M1][3], d_ccw [N][M1], d_nraw [N][4] overwritten (d_nraw by the slot-0 thread; column 3 = 0: the row is the noise MLP's
// d_raw as it stands).
__global__ void blur_mix_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb0, const float* __restrict__ ccw,
                                    const float* __restrict__ nraw, int nraw_ld, int N, int M1, int gamma, const float* __restrict__ g_blur,
                                    const float* __restrict__ g_blur0, const float* __restrict__ g_noise,
                                    const float* __restrict__ g_sharp, const float* __restrict__ g_sharp0,
                                    float* __restrict__ d_rgb, float* __restrict__ d_rgb0, float* __restrict__ d_ccw,
                                    float* __restrict__ d_nraw) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * M1) return;
    const long long n = t / M1;
    const int slot = (int)(t % M1);
    const float w = ccw[t];
    float dw = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f, s0 = 0.f;            // (recomputed per thread: M1 * 6 loads that the wave's neighbours share through L1)
        for (int m = 0; m < M1; ++m) {
            const float wm = ccw[n * M1 + m];
            s += rgb[(n * M1 + m) * 3 + c] * wm;
            s0 += rgb0[(n * M1 + m) * 3 + c] * wm;
        }
        const float sg = 1.f / (1.f + expf(-nraw[n * nraw_ld + c]));
        const float nz = 0.1f * sg;
        const float gb = g_blur ? g_blur[n * 3 + c] * tm_slope(s + nz, gamma) : 0.f;
        const float gb0 = g_blur0 ? g_blur0[n * 3 + c] * tm_slope(s0 + nz, gamma) : 0.f;
        const float gs = gb + (g_sharp ? g_sharp[n * 3 + c] * tm_slope(s, gamma) : 0.f);       // d / d (weighted sum), fine
        const float gs0 = gb0 + (g_sharp0 ? g_sharp0[n * 3 + c] * tm_slope(s0, gamma) : 0.f);  // ... coarse
        d_rgb[t * 3 + c] = gs * w;
        d_rgb0[t * 3 + c] = gs0 * w;
        dw += gs * rgb[t * 3 + c] + gs0 * rgb0[t * 3 + c];
        if (slot == 0) d_nraw[n * 4 + c] = (gb + gb0 + (g_noise ? g_noise[n * 3 + c] : 0.f)) * 0.1f * sg * (1.f - sg);
    }
    if (slot == 0) d_nraw[n * 4 + 3] = 0.f;
    d_ccw[t] = dw;
}

// work == NULL: loss[0] is added to (the caller zeroed it).  work != NULL ({running sum, finished workgroups}, both zero on
// entry and left zero): loss[0] is WRITTEN by the last workgroup to finish -- no zero-fill launch in front (ABI 7).
__global__ void loss_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ tg,
                            int n3, float scale, float* __restrict__ loss, float* __restrict__ ga, float* __restrict__ gb,
                            float* __restrict__ work) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    if (t < n3) {
        const float inv = scale / (float)n3;
        const float da = a[t] - tg[t], db = b[t] - tg[t];
        l = 0.5f * (da * da + fabsf(da) + db * db + fabsf(db)) * inv;
        const float va = (da + 0.5f * (da > 0.f ? 1.f : (da < 0.f ? -1.f : 0.f))) * inv;
        const float vb = (db + 0.5f * (db > 0.f ? 1.f : (db < 0.f ? -1.f : 0.f))) * inv;
        if (gb != nullptr) { ga[t] = va; gb[t] = vb; }
        else ga[t] = va + vb;           // a and b are the same tensor (no fine pass: rgb0 = rgb): its gradient is the sum
    }
    l = wave_sum(l);
    if (work == nullptr) {
        if ((threadIdx.x & 63) == 0 && l != 0.f) atomicAdd(loss, l);
        return;
    }
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(work, part[0] + part[1] + part[2] + part[3]);
        __threadfence();
        if (atomicAdd(reinterpret_cast<unsigned*>(work) + 1, 1u) == gridDim.x - 1) {
            loss[0] = atomicExch(work, 0.f);
            reinterpret_cast<unsigned*>(work)[1] = 0u;
        }
    }
}

// ------------------------------------------------------------------ random draws
// The four draws of one march (models/lushnerf.py:515 torch.rand [R,Ns]; :322 torch.randn_like [R,Ns-1]; helpers:578
// torch.rand [R,Ni]; :322 [R,Ns+Ni-1]) in ONE launch: Philox4x32-10 counter RNG, key = (seed, stream), counter = (element
// quad, array id, call offset).  Uniforms are 24-bit in [0, 1) as torch.rand's; normals are Box-Muller pairs.  The
// reference's own torch RNG stream cannot be reproduced (it differs between CPU and GPU builds of torch itself);
// parity tests pass explicit draws instead.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
struct DrawArr { float* p; long long n; int normal; };
struct DrawSet { DrawArr a[4]; };
__global__ void draws_kernel(DrawSet S, unsigned long long seed, unsigned long long offset, long long total_quads,
                             const unsigned long long* __restrict__ base_dev) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total_quads) return;
    if (base_dev) offset += *base_dev;          // the step state's draw counter (lush_step_state): a captured launch still advances
    long long base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long nq = (S.a[i].n + 3) / 4;
        if (q >= base && q < base + nq) {
            const long long e = q - base;
            unsigned c[4] = {(unsigned)e, (unsigned)(e >> 32), (unsigned)i, (unsigned)offset};
            philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32) ^ (unsigned)(offset >> 32));
            float v[4];
            if (S.a[i].normal) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float u1 = ((float)(c[2 * h] >> 8) + 1.0f) * 5.9604644775390625e-08f;          // (0, 1]
                    const float u2 = (float)(c[2 * h + 1] >> 8) * 5.9604644775390625e-08f;                // [0, 1)
                    const float rad = sqrtf(-2.0f * logf(u1));
                    float sn, cs;
                    sincosf(6.283185307179586f * u2, &sn, &cs);
                    v[2 * h] = rad * cs;
                    v[2 * h + 1] = rad * sn;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; ++h) v[h] = (float)(c[h] >> 8) * 5.9604644775390625e-08f;          // [0, 1), 24 bits
            }
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (4 * e + h < S.a[i].n) S.a[i].p[4 * e + h] = v[h];
        }
        base += nq;
    }
}

// ------------------------------------------------------- d rays from d points
__global__ __launch_bounds__(RAYS_PER_BLOCK * 64) void ray_grad_reduce_kernel(const float* __restrict__ dpts,
        const float* __restrict__ z, int R, int S, float* __restrict__ drays) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= R) return;
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (up to 256 samples with every load in flight before the first is used: as a loop with a run-time trip count each 64
    // samples waited for their own three loads; more samples take the loop)
    for (int s0 = 0; s0 < S; s0 += 256) {
        float4 gx[4], gv[4];
        float zz[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int s = s0 + c * 64 + lane;
            const bool in = s < S;
            const long long p = (long long)ray * S + (in ? s : 0);
            gx[c] = *reinterpret_cast<const float4*>(dpts + p * 8);
            gv[c] = *reinterpret_cast<const float4*>(dpts + p * 8 + 4);
            zz[c] = z[p];
            if (!in) { gx[c] = make_float4(0.f, 0.f, 0.f, 0.f); gv[c] = gx[c]; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {          // (chunks in order: the order of the sums is the loop's)
            if (s0 + c * 64 < S) {
                a[0] += gx[c].x; a[1] += gx[c].y; a[2] += gx[c].z;
                a[3] += gx[c].x * zz[c]; a[4] += gx[c].y * zz[c]; a[5] += gx[c].z * zz[c];
                a[6] += gv[c].x; a[7] += gv[c].y; a[8] += gv[c].z;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = wave_sum(a[i]);
    // one addition per word by lanes 0..8 (a read-modify-write by lane 0 waited for its own nine loads; this ray's words are
    // touched by no other wave of the launch, and the launches of a step run one after the other)
    float mine = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) mine = lane == i ? a[i] : mine;
    if (lane < 9) atomicAdd(drays + (long long)ray * 11 + (lane < 6 ? lane : lane + 2), mine);
}

// ------------------------------------------------------- live points (round 5)
// A sample whose density pre-activation is clamped by the ReLU of raw2outputs (models/lushnerf.py:313: relu(raw + noise)) has
// alpha = 0, weight = 0 and d alpha / d raw = 0: its d_raw row is EXACTLY zero and nothing flows back through its MLP
// evaluation.  With raw_noise_std = 1 (every shipped config) and a density near zero that is half of all the points.  The
// backward therefore runs on the LIVE points only: this compaction lists them (in grid order: deterministic), gathers their
// d_raw rows and gives every ray its range of the list; the forward is re-run with the stash on that list, the gradient
// chain and the weight gradients see a dense launch of `cnt` points.  Sums skip exact zeros only: same gradients.
constexpr int LIVE_BLK = 1024;
__device__ __forceinline__ bool live_row(const float4 v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; }      // (NaN counts as live)

__global__ __launch_bounds__(256) void live_count_kernel(const float4* __restrict__ draw, int P, int* __restrict__ blk) {
    const int p0 = blockIdx.x * LIVE_BLK + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (p0 + k < P) c += live_row(draw[p0 + k]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __shared__ int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// one workgroup: exclusive scan of the block counts in place; cnt[0] = the number of live points, cnt[1] = P
__global__ __launch_bounds__(1024) void live_scan_kernel(int* __restrict__ blk, int nblk, int P, int* __restrict__ cnt) {
    __shared__ int sm[1024];
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int i = b0 + (int)threadIdx.x;
        const int v = i < nblk ? blk[i] : 0;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
            __syncthreads();
            sm[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nblk) blk[i] = carry + sm[threadIdx.x] - v;
        const int tot = sm[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) { cnt[0] = carry; cnt[1] = P; }
}
__global__ __launch_bounds__(256) void live_scatter_kernel(const float4* __restrict__ draw, int P, int S, int R, const int* __restrict__ blk_off,
                                                          const int* __restrict__ cnt, int* __restrict__ live_idx,
                                                          float4* __restrict__ draw_c, int* __restrict__ ray_start) {
    const int p0 = blockIdx.x * LIVE_BLK + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 v[4];
    int f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = p0 + k < P ? draw[p0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        f[k] = live_row(v[k]) ? 1 : 0;
        c += f[k];
    }
    int incl = c;                                   // inclusive scan over the workgroup's 256 threads
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(incl, o, 64); if (lane >= o) incl += a; }
    __shared__ int sm[4];
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    int base = blk_off[blockIdx.x];
    for (int i = 0; i < w; ++i) base += sm[i];
    int pos = base + incl - c;
    const int total = cnt[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k;
        if (p < P) {
            if (p % S == 0) ray_start[p / S] = pos;          // where ray p / S begins in the list
            if (f[k]) { live_idx[pos] = p; draw_c[pos] = v[k]; ++pos; }
            if (p >= total) draw_c[p] = make_float4(0.f, 0.f, 0.f, 0.f);      // rows behind the list: zero (tile padding, the loss scale's pass)
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ray_start[R] = total;
}

// d rays from the d(point) rows of a live-point launch: ray r owns rows [ray_start[r], ray_start[r + 1]) of the list
__global__ __launch_bounds__(RAYS_PER_BLOCK * 64) void ray_grad_reduce_live_kernel(const float* __restrict__ dpts, const float* __restrict__ z,
        const int* __restrict__ live_idx, const int* __restrict__ ray_start, int R, float* __restrict__ drays) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int b = ray_start[ray], e = ray_start[ray + 1];
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = b + lane; i < e; i += 64) {
        const float4 gx = *reinterpret_cast<const float4*>(dpts + (long long)i * 8);
        const float4 gv = *reinterpret_cast<const float4*>(dpts + (long long)i * 8 + 4);
        const float zz = z[live_idx[i]];
        a[0] += gx.x; a[1] += gx.y; a[2] += gx.z;
        a[3] += gx.x * zz; a[4] += gx.y * zz; a[5] += gx.z * zz;
        a[6] += gv.x; a[7] += gv.y; a[8] += gv.z;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = wave_sum(a[i]);
    float mine = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) mine = lane == i ? a[i] : mine;
    if (lane < 9) atomicAdd(drays + (long long)ray * 11 + (lane < 6 ? lane : lane + 2), mine);
}

// ------------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // torch: exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// Step state on the device (include/lush_march.h lush_step_state): what a training step takes from the host as kernel
// arguments otherwise -- learning rate, Adam's step counts (as their bias corrections), the Philox draw counter -- so that
// a step captured in a HIP graph advances when it is replayed.
struct StepState {
    unsigned long long draw_base;   // draw calls made before this step
    int global_step;                // steps taken
    int steps[3];                   // Adam steps taken per segment
    float lr;                       // rate of the next step: lrate * 0.1 ** (max(global_step - 1, 0) / decay_steps)
    float bc1[3], bc2s[3];          // 1 - beta1^t, sqrt(1 - beta2^t) of the next step (t = steps + 1) per segment
};
__global__ void adam_state_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long long n, const StepState* __restrict__ st, int seg, float b1,
                                  float b2, float eps, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lr = st->lr, bc1 = st->bc1[seg], bc2_sqrt = st->bc2s[seg];
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
// the same update over SEVERAL segments of one flat buffer in one launch: element i belongs to segment (i >= end0) + (i >= end1);
// segments whose bit is clear in `mask` are left alone (the reference's grad=None parameters: Adam skips them, counts included)
__global__ void adam_state_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                        float* __restrict__ v, long long end0, long long end1, long long end2,
                                        const StepState* __restrict__ st, int mask, float b1, float b2, float eps, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end2) return;
    const int seg = (i >= end0) + (i >= end1);
    if (!((mask >> seg) & 1)) return;
    const float lr = st->lr, bc1 = st->bc1[seg], bc2_sqrt = st->bc2s[seg];
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
struct AdamMulti { long long end[3]; float bc1[3], bc2s[3]; int mask; };
__global__ void adam_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                  const AdamMulti A, float lr, float b1, float b2, float eps, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.end[2]) return;
    const int seg = (i >= A.end[0]) + (i >= A.end[1]);
    if (!((A.mask >> seg) & 1)) return;
    const float bc1 = seg == 0 ? A.bc1[0] : seg == 1 ? A.bc1[1] : A.bc1[2];
    const float bc2_sqrt = seg == 0 ? A.bc2s[0] : seg == 1 ? A.bc2s[1] : A.bc2s[2];
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
__device__ void step_state_derive(StepState* st, double lrate, double decay_steps, double b1, double b2) {
    const int g = st->global_step - 1 > 0 ? st->global_step - 1 : 0;
    st->lr = (float)(lrate * pow(0.1, (double)g / decay_steps));
    for (int s = 0; s < 3; ++s) {
        const double t = (double)(st->steps[s] + 1);
        st->bc1[s] = (float)(1.0 - pow(b1, t));
        st->bc2s[s] = (float)sqrt(1.0 - pow(b2, t));
    }
}
__global__ void step_state_advance_kernel(StepState* st, int n_draw_calls, int active_mask, double lrate, double decay_steps,
                                          double b1, double b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->draw_base += (unsigned long long)n_draw_calls;
    for (int s = 0; s < 3; ++s)
        if ((active_mask >> s) & 1) st->steps[s] += 1;
    st->global_step += 1;
    step_state_derive(st, lrate, decay_steps, b1, b2);
}
__global__ void step_state_init_kernel(StepState* st, unsigned long long draw_base, int global_step, int s0, int s1, int s2,
                                       double lrate, double decay_steps, double b1, double b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->draw_base = draw_base; st->global_step = global_step;
    st->steps[0] = s0; st->steps[1] = s1; st->steps[2] = s2;
    step_state_derive(st, lrate, decay_steps, b1, b2);
}

}  // namespace lush

// =============================================================================
// C ABI
// =============================================================================
using namespace lush;
#define S_(s) ((hipStream_t)(s))
#define CHECK_LAUNCH() LUSH_HIP(hipGetLastError())
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

extern "C" {

const char* lush_last_error(void) { return g_err.c_str(); }
int lush_abi_version(void) { return 10; }

int lush_zgrid(const float* rays, int R, int S, int lindisp, const float* t_rand, float* z, lush_stream_t st) {
    if (R <= 0 || S <= 0) return set_error("lush_zgrid: empty");
    hipLaunchKernelGGL(zgrid_kernel, dim3(cdiv((long long)R * S, 256)), dim3(256), 0, S_(st), rays, R, S, lindisp, t_rand, z);
    CHECK_LAUNCH();
    return 0;
}
int lush_zfixed(const float* rays, int R, int S, int index, int lindisp, float* z, lush_stream_t st) {
    if (index < 0 || index >= S) return set_error("lush_zfixed: index out of range");
    hipLaunchKernelGGL(zfixed_kernel, dim3(cdiv(R, 256)), dim3(256), 0, S_(st), rays, R, S, index, lindisp, z);
    CHECK_LAUNCH();
    return 0;
}

int lush_composite_fwd(const float* raw, const float* z, const float* rays, int R, int S, const float* noise,
                       float noise_std, float near_mask, int white_bkgd, float* rgb, float* depth, float* acc,
                       float* weights, float* density, int* flags, int flag_shift, lush_stream_t st) {
    if (S < 2 || S > 256) return set_error("lush_composite_fwd: S must be in [2,256]");
    if (flag_shift < 0 || flag_shift > 16) return set_error("lush_composite_fwd: flag_shift must be in [0,16]");
    CompIn c{raw, z, rays, noise, R, S, noise_std, near_mask, white_bkgd};
    dim3 g(cdiv(R, RAYS_PER_BLOCK)), b(RAYS_PER_BLOCK * 64);
    if (S <= 64) hipLaunchKernelGGL(composite_fwd_kernel<1>, g, b, 0, S_(st), c, rgb, depth, acc, weights, density, flags, flag_shift);
    else if (S <= 128) hipLaunchKernelGGL(composite_fwd_kernel<2>, g, b, 0, S_(st), c, rgb, depth, acc, weights, density, flags, flag_shift);
    else hipLaunchKernelGGL(composite_fwd_kernel<4>, g, b, 0, S_(st), c, rgb, depth, acc, weights, density, flags, flag_shift);
    CHECK_LAUNCH();
    return 0;
}
static int composite_bwd_blocks(int R) {
    // at most 2048 workgroups (8 per CU: what is resident at once) walk the rays
    int blocks = cdiv(R, RAYS_PER_BLOCK);
    return blocks > 2048 ? 2048 : blocks;
}
int lush_composite_bwd_blocks(int R) { return R > 0 ? composite_bwd_blocks(R) : 0; }

int lush_composite_bwd(const float* raw, const float* z, const float* rays, int R, int S, const float* noise,
                       float noise_std, float near_mask, int white_bkgd, const float* g_rgb, const float* g_depth,
                       const float* g_acc, float* draw, float* drays, float* block_max, float* zero_buf, long long zero_n,
                       int init_drays, lush_stream_t st) {
    if (S < 2 || S > 256) return set_error("lush_composite_bwd: S must be in [2,256]");
    if (zero_buf != nullptr && zero_n < 0) return set_error("lush_composite_bwd: zero_n must not be negative");
    CompIn c{raw, z, rays, noise, R, S, noise_std, near_mask, white_bkgd};
    const CompBwdExtra x{block_max, zero_buf, zero_buf ? zero_n : 0, init_drays};
    dim3 g(composite_bwd_blocks(R)), b(RAYS_PER_BLOCK * 64);
    if (S <= 64) hipLaunchKernelGGL(composite_bwd_kernel<1>, g, b, 0, S_(st), c, g_rgb, g_depth, g_acc, draw, drays, x);
    else if (S <= 128) hipLaunchKernelGGL(composite_bwd_kernel<2>, g, b, 0, S_(st), c, g_rgb, g_depth, g_acc, draw, drays, x);
    else hipLaunchKernelGGL(composite_bwd_kernel<4>, g, b, 0, S_(st), c, g_rgb, g_depth, g_acc, draw, drays, x);
    CHECK_LAUNCH();
    return 0;
}
int lush_loss_scale(const float* block_max, int n, float* scale2, lush_stream_t st) {
    if (!block_max || n < 1 || !scale2) return set_error("lush_loss_scale: maxima and destination are required");
    hipLaunchKernelGGL(loss_scale_kernel, dim3(1), dim3(256), 0, S_(st), block_max, n, scale2);
    CHECK_LAUNCH();
    return 0;
}

int lush_sample_merge(const float* z, const float* weights, int R, int S, int Ni, const float* u, float* z_out,
                      float* z_samples, float* z_std, int* flags, lush_stream_t st) {
    if (S < 3 || S > 256 || Ni < 1 || S + Ni > SM_MAXN) return set_error("lush_sample_merge: need 3<=S<=256, S+Ni<=512");
    hipLaunchKernelGGL(sample_merge_kernel, dim3(cdiv(R, RAYS_PER_BLOCK)), dim3(RAYS_PER_BLOCK * 64), 0, S_(st), z,
                       weights, R, S, Ni, u, z_out, z_samples, z_std, flags);
    CHECK_LAUNCH();
    return 0;
}

int lush_pack_rays_fwd(const float* rays, int N, int ndc, float cx, float cy, float near, float far, float* batch,
                       lush_stream_t st) {
    hipLaunchKernelGGL(pack_rays_fwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, S_(st), rays, N, ndc, cx, cy, near, far, batch);
    CHECK_LAUNCH();
    return 0;
}
int lush_pack_rays_bwd(const float* rays, int N, int ndc, float cx, float cy, const float* dbatch, float* drays,
                       lush_stream_t st) {
    hipLaunchKernelGGL(pack_rays_bwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, S_(st), rays, N, ndc, cx, cy, dbatch, drays);
    CHECK_LAUNCH();
    return 0;
}

int lush_gen_rays(const float* c2w, const int64_t* view, const int64_t* px, const int64_t* py, int N, float fx,
                  float fy, float cx, float cy, float* rays, lush_stream_t st) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(gen_rays_kernel, dim3(cdiv(N, 256)), dim3(256), 0, S_(st), c2w, view, px, py, N, fx, fy, cx, cy, rays);
    CHECK_LAUNCH();
    return 0;
}

int lush_gen_rays_image(const float* c2w, int H, int W, float fx, float fy, float cx, float cy, float* rays,
                        lush_stream_t st) {
    if (H <= 0 || W <= 0) return set_error("lush_gen_rays_image: empty image");
    hipLaunchKernelGGL(gen_rays_image_kernel, dim3(cdiv((long long)H * W, 256)), dim3(256), 0, S_(st), c2w, H, W, fx, fy, cx, cy, rays);
    CHECK_LAUNCH();
    return 0;
}

int lush_align_rays(const float* c2w, const float* align, const void* cert, int cert_is_u8, const int64_t* samples,
                    int V, int ns, long long HW, int H, int W, float fx, float fy, float cx, float cy, float* rays,
                    float* cert_out, lush_stream_t st) {
    if (V <= 0 || ns <= 0) return set_error("lush_align_rays: empty");
    if (HW <= 0) return set_error("lush_align_rays: empty match table");
    hipLaunchKernelGGL(align_rays_kernel, dim3(cdiv((long long)V * ns, 128)), dim3(128), 0, S_(st), c2w, align, cert, cert_is_u8,
                       samples, V, ns, HW, H, W, fx, fy, cx, cy, rays, cert_out);
    CHECK_LAUNCH();
    return 0;
}

int lush_consist_loss_fwd_bwd(const float* rgb, const float* cert, int V, int ns, float threshold, float* loss,
                              float* grad, lush_stream_t st) {
    if (V <= 0 || ns <= 0) return set_error("lush_consist_loss_fwd_bwd: empty");
    hipLaunchKernelGGL(consist_loss_kernel, dim3(1), dim3(256), 0, S_(st), rgb, cert, V, ns, threshold, loss, grad);
    CHECK_LAUNCH();
    return 0;
}

int lush_rbk_mlp_fwd(const lush_rbk_params* p, int num_img, int M, float window, float* acts, lush_stream_t st) {
    if (M < 1 || M > 4) return set_error("lush_rbk_mlp_fwd: 1 <= num_motion <= 4");
    if (num_img < 1) return set_error("lush_rbk_mlp_fwd: no images");
    const size_t lds = ((size_t)RBK_IPB * RBK_LS + RBK_WALL) * sizeof(float);
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rbk_mlp_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rbk_mlp_fwd_kernel, dim3(cdiv(num_img, RBK_IPB)), dim3(RBK_NT), lds, S_(st), *p, num_img, M, window, acts);
    CHECK_LAUNCH();
    return 0;
}
int lush_rbk_mlp_bwd(const lush_rbk_params* p, int num_img, int M, float window, const float* acts,
                     float* d_rvw, int rvw_stride, const lush_rbk_grads* g, float* scratch, int accumulate, lush_stream_t st) {
    if (M < 1 || M > 4) return set_error("lush_rbk_mlp_bwd: 1 <= num_motion <= 4");
    if (num_img < 1) return set_error("lush_rbk_mlp_bwd: no images");
    if (rvw_stride < LUSH_RBK_RVW_STRIDE) return set_error("lush_rbk_mlp_bwd: rvw_stride must be at least LUSH_RBK_RVW_STRIDE");
    // the weight gradients are summed by atomics into buffers that hold values -- the trainer's flat gradient (accumulate != 0),
    // or zeros written here first (accumulate == 0: 19 small memsets, not a path a training step takes)
    if (!accumulate) {
        const size_t f = sizeof(float);
        LUSH_HIP(hipMemsetAsync(g->embed, 0, (size_t)num_img * 64 * f, S_(st)));
        for (int l = 0; l < 4; ++l) {
            LUSH_HIP(hipMemsetAsync(g->w_trunk[l], 0, 64 * 64 * f, S_(st)));
            LUSH_HIP(hipMemsetAsync(g->b_trunk[l], 0, 64 * f, S_(st)));
        }
        float* w32[3] = {g->w_rb, g->w_vb, g->w_wb};
        float* b32[3] = {g->b_rb, g->b_vb, g->b_wb};
        for (int i = 0; i < 3; ++i) {
            LUSH_HIP(hipMemsetAsync(w32[i], 0, 32 * 64 * f, S_(st)));
            LUSH_HIP(hipMemsetAsync(b32[i], 0, 32 * f, S_(st)));
        }
        LUSH_HIP(hipMemsetAsync(g->w_r, 0, (size_t)3 * M * 32 * f, S_(st))); LUSH_HIP(hipMemsetAsync(g->b_r, 0, (size_t)3 * M * f, S_(st)));
        LUSH_HIP(hipMemsetAsync(g->w_v, 0, (size_t)3 * M * 32 * f, S_(st))); LUSH_HIP(hipMemsetAsync(g->b_v, 0, (size_t)3 * M * f, S_(st)));
        LUSH_HIP(hipMemsetAsync(g->w_w, 0, (size_t)(M + 1) * 32 * f, S_(st))); LUSH_HIP(hipMemsetAsync(g->b_w, 0, (size_t)(M + 1) * f, S_(st)));
    }
    (void)scratch;      // (unused since the LDS version; kept in the signature)
    const size_t lds = ((size_t)2 * RBK_IPB * RBK_LS + RBK_WALL) * sizeof(float);
    LUSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rbk_mlp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rbk_mlp_bwd_kernel, dim3(cdiv(num_img, RBK_IPB)), dim3(RBK_NT), lds, S_(st), *p, num_img, M, window, acts, d_rvw, *g,
                       rvw_stride);
    CHECK_LAUNCH();
    return 0;
}
int lush_rbk_warp_fwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, float* new_rays,
                      float* ccw, lush_stream_t st) {
    hipLaunchKernelGGL(rbk_warp_fwd_kernel, dim3(cdiv(N, 128)), dim3(128), 0, S_(st), rays, idx, N, M, acts, new_rays, ccw);
    CHECK_LAUNCH();
    return 0;
}
int lush_rbk_warp_bwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, const float* dnew_rays,
                      const float* dccw, const uint8_t* mask, float* d_rvw, int rvw_stride, float* drays, lush_stream_t st) {
    if (rvw_stride < LUSH_RBK_RVW_STRIDE) return set_error("lush_rbk_warp_bwd: rvw_stride must be at least LUSH_RBK_RVW_STRIDE");
    hipLaunchKernelGGL(rbk_warp_bwd_kernel, dim3(cdiv(N, 128)), dim3(128), 0, S_(st), rays, idx, N, M, acts, dnew_rays, dccw, mask, d_rvw, rvw_stride, drays);
    CHECK_LAUNCH();
    return 0;
}

int lush_rbk_warp_ndc_fwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, int ndc, float cx, float cy,
                          float near, float far, float* batch, float* ccw, float* batch0, lush_stream_t st) {
    if (N < 1 || M < 1 || M > 4) return set_error("lush_rbk_warp_ndc_fwd: need N >= 1 and 1 <= num_motion <= 4");
    if (!rays || !idx || !acts || !batch || !ccw) return set_error("lush_rbk_warp_ndc_fwd: rays, idx, acts, batch and ccw are required");
    hipLaunchKernelGGL(rbk_warp_ndc_fwd_kernel, dim3(cdiv((long long)N * (M + 1), 256)), dim3(256), 0, S_(st), rays, idx, N, M, acts, ndc,
                       cx, cy, near, far, batch, ccw, batch0);
    CHECK_LAUNCH();
    return 0;
}
int lush_rbk_warp_ndc_bwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, int ndc, float cx, float cy,
                          const float* dbatch, const float* dccw, const uint8_t* mask, float* d_rvw, int rvw_stride, float* drays,
                          int num_img, lush_stream_t st) {
    if (N < 1 || M < 1 || M > 4) return set_error("lush_rbk_warp_ndc_bwd: need N >= 1 and 1 <= num_motion <= 4");
    if (!rays || !idx || !acts || !d_rvw) return set_error("lush_rbk_warp_ndc_bwd: rays, idx, acts and d_rvw are required");
    if (rvw_stride < LUSH_RBK_RVW_STRIDE) return set_error("lush_rbk_warp_ndc_bwd: rvw_stride must be at least LUSH_RBK_RVW_STRIDE");
    if (num_img < 1) return set_error("lush_rbk_warp_ndc_bwd: num_img (the number of rows of acts / d_rvw) must be given");
    const int M1 = M + 1;
    int rpb = 256 / M1;                        // ~256 threads per workgroup: enough workgroups to spread over the chip
    if (rpb > N) rpb = N;
    hipLaunchKernelGGL(rbk_warp_ndc_bwd_kernel, dim3(cdiv(N, rpb)), dim3(rpb * M1), 0, S_(st), rays, idx, N, M, acts, ndc, cx, cy, dbatch,
                       dccw, mask, d_rvw, rvw_stride, drays, num_img, rpb);
    CHECK_LAUNCH();
    return 0;
}
int lush_blur_mix_fwd(const float* rgb, const float* rgb0, const float* ccw, const float* nraw, int nraw_ld, int N, int M1, int gamma,
                      float* blur, float* blur0, float* noise, float* sharp, float* sharp0, lush_stream_t st) {
    if (N < 1 || M1 < 1 || nraw_ld < 3) return set_error("lush_blur_mix_fwd: empty, or a row stride below 3");
    if (!rgb || !rgb0 || !ccw || !nraw || !blur || !blur0 || !noise || !sharp || !sharp0) return set_error("lush_blur_mix_fwd: every pointer is required");
    hipLaunchKernelGGL(blur_mix_fwd_kernel, dim3(cdiv(3LL * N, 256)), dim3(256), 0, S_(st), rgb, rgb0, ccw, nraw, nraw_ld, N, M1, gamma, blur, blur0,
                       noise, sharp, sharp0);
    CHECK_LAUNCH();
    return 0;
}
int lush_blur_mix_bwd(const float* rgb, const float* rgb0, const float* ccw, const float* nraw, int nraw_ld, int N, int M1, int gamma,
                      const float* g_blur, const float* g_blur0, const float* g_noise, const float* g_sharp, const float* g_sharp0,
                      float* d_rgb, float* d_rgb0, float* d_ccw, float* d_nraw, lush_stream_t st) {
    if (N < 1 || M1 < 1 || nraw_ld < 3) return set_error("lush_blur_mix_bwd: empty, or a row stride below 3");
    if (!rgb || !rgb0 || !ccw || !nraw || !d_rgb || !d_rgb0 || !d_ccw || !d_nraw) return set_error("lush_blur_mix_bwd: inputs and the four gradient outputs are required");
    hipLaunchKernelGGL(blur_mix_bwd_kernel, dim3(cdiv((long long)N * M1, 256)), dim3(256), 0, S_(st), rgb, rgb0, ccw, nraw, nraw_ld, N, M1, gamma,
                       g_blur, g_blur0, g_noise, g_sharp, g_sharp0, d_rgb, d_rgb0, d_ccw, d_nraw);
    CHECK_LAUNCH();
    return 0;
}

int lush_wsum_fwd(const float* x, const float* ccw, int N, int M, int C, float* y, lush_stream_t st) {
    hipLaunchKernelGGL(wsum_fwd_kernel, dim3(cdiv((long long)N * C, 256)), dim3(256), 0, S_(st), x, ccw, N, M, C, y);
    CHECK_LAUNCH();
    return 0;
}
int lush_wsum_bwd(const float* x, const float* ccw, int N, int M, int C, const float* dy, float* dx, float* dccw,
                  lush_stream_t st) {
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(cdiv((long long)N * M, 256)), dim3(256), 0, S_(st), x, ccw, N, M, C, dy, dx, dccw);
    CHECK_LAUNCH();
    return 0;
}
int lush_tonemap_fwd(const float* x, const float* nraw, int n, int gamma, float* y, lush_stream_t st) {
    hipLaunchKernelGGL(tonemap_fwd_kernel, dim3(cdiv(3LL * n, 256)), dim3(256), 0, S_(st), x, nraw, 3 * n, gamma, y);
    CHECK_LAUNCH();
    return 0;
}
int lush_tonemap_bwd(const float* x, const float* nraw, int n, int gamma, const float* dy, float* dx, float* dnraw,
                     lush_stream_t st) {
    hipLaunchKernelGGL(tonemap_bwd_kernel, dim3(cdiv(3LL * n, 256)), dim3(256), 0, S_(st), x, nraw, 3 * n, gamma, dy, dx, dnraw);
    CHECK_LAUNCH();
    return 0;
}
int lush_noise_act_fwd(const float* x, int n, float* y, lush_stream_t st) {
    hipLaunchKernelGGL(noise_act_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(st), x, n, y);
    CHECK_LAUNCH();
    return 0;
}
int lush_noise_act_bwd(const float* x, int n, const float* dy, float* dx, lush_stream_t st) {
    hipLaunchKernelGGL(noise_act_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(st), x, n, dy, dx);
    CHECK_LAUNCH();
    return 0;
}
int lush_loss_fwd_bwd(const float* a, const float* b, const float* target, int n, float scale, float* loss, float* ga,
                      float* gb, float* work, lush_stream_t st) {
    if (!gb && a != b) return set_error("lush_loss_fwd_bwd: one gradient buffer serves only a == b");
    hipLaunchKernelGGL(loss_kernel, dim3(cdiv(3LL * n, 256)), dim3(256), 0, S_(st), a, b, target, 3 * n, scale, loss, ga, gb, work);
    CHECK_LAUNCH();
    return 0;
}

static int draws_impl(unsigned long long seed, unsigned long long offset, const unsigned long long* base_dev, float* t_rand,
                      long long n_t, float* noise_c, long long n_c, float* u, long long n_u, float* noise_f, long long n_f,
                      lush_stream_t st) {
    DrawSet S{};
    S.a[0] = {t_rand, t_rand ? n_t : 0, 0};
    S.a[1] = {noise_c, noise_c ? n_c : 0, 1};
    S.a[2] = {u, u ? n_u : 0, 0};
    S.a[3] = {noise_f, noise_f ? n_f : 0, 1};
    long long quads = 0;
    for (int i = 0; i < 4; ++i) quads += (S.a[i].n + 3) / 4;
    if (quads == 0) return 0;
    hipLaunchKernelGGL(draws_kernel, dim3(cdiv(quads, 256)), dim3(256), 0, S_(st), S, seed, offset, quads, base_dev);
    CHECK_LAUNCH();
    return 0;
}
int lush_draws(unsigned long long seed, unsigned long long offset, float* t_rand, long long n_t, float* noise_c,
               long long n_c, float* u, long long n_u, float* noise_f, long long n_f, lush_stream_t st) {
    return draws_impl(seed, offset, nullptr, t_rand, n_t, noise_c, n_c, u, n_u, noise_f, n_f, st);
}
int lush_draws_state(unsigned long long seed, unsigned long long offset, const void* state, float* t_rand, long long n_t,
                     float* noise_c, long long n_c, float* u, long long n_u, float* noise_f, long long n_f, lush_stream_t st) {
    if (!state) return set_error("lush_draws_state: the step state is required");
    return draws_impl(seed, offset, &static_cast<const StepState*>(state)->draw_base, t_rand, n_t, noise_c, n_c, u, n_u, noise_f, n_f, st);
}

int lush_ray_grad_reduce(const float* dpts, const float* z, int R, int S, float* drays, lush_stream_t st) {
    hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3(cdiv(R, RAYS_PER_BLOCK)), dim3(RAYS_PER_BLOCK * 64), 0, S_(st), dpts, z, R, S, drays);
    CHECK_LAUNCH();
    return 0;
}

size_t lush_live_aux_bytes(long long P) { return P > 0 ? (size_t)((P + LIVE_BLK - 1) / LIVE_BLK) * sizeof(int) : 0; }

int lush_live_compact(const float* draw, int R, int S, int* live_idx, float* draw_c, int* ray_start, int* cnt, void* aux, lush_stream_t st) {
    const long long P = (long long)R * S;
    if (R <= 0 || S <= 0 || P >= (1LL << 27)) return set_error("lush_live_compact: 1 .. 2^27 - 1 points");
    if (!draw || !live_idx || !draw_c || !ray_start || !cnt || !aux) return set_error("lush_live_compact: every buffer is required");
    const int nblk = (int)((P + LIVE_BLK - 1) / LIVE_BLK);
    int* blk = (int*)aux;
    hipLaunchKernelGGL(live_count_kernel, dim3(nblk), dim3(256), 0, S_(st), (const float4*)draw, (int)P, blk);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(live_scan_kernel, dim3(1), dim3(1024), 0, S_(st), blk, nblk, (int)P, cnt);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(live_scatter_kernel, dim3(nblk), dim3(256), 0, S_(st), (const float4*)draw, (int)P, S, R, (const int*)blk, (const int*)cnt,
                       live_idx, (float4*)draw_c, ray_start);
    CHECK_LAUNCH();
    return 0;
}

int lush_ray_grad_reduce_live(const float* dpts, const float* z, const int* live_idx, const int* ray_start, int R, float* drays, lush_stream_t st) {
    hipLaunchKernelGGL(ray_grad_reduce_live_kernel, dim3(cdiv(R, RAYS_PER_BLOCK)), dim3(RAYS_PER_BLOCK * 64), 0, S_(st), dpts, z, live_idx, ray_start, R, drays);
    CHECK_LAUNCH();
    return 0;
}

int lush_adam(float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1, float beta2,
              float eps, int step, float grad_scale, lush_stream_t st) {
    if (n <= 0) return 0;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(st), param, grad, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
    CHECK_LAUNCH();
    return 0;
}

int lush_adam_multi(float* param, const float* grad, float* m, float* v, long long end0, long long end1, long long end2, int mask,
                    float lr, float beta1, float beta2, float eps, const int* steps, float grad_scale, lush_stream_t st) {
    if (!steps || end0 < 0 || end1 < end0 || end2 < end1 || mask < 0 || mask > 7) return set_error("lush_adam_multi: bad segment ends / mask / steps");
    AdamMulti A;
    A.end[0] = end0; A.end[1] = end1;
    A.end[2] = (mask & 4) ? end2 : (mask & 2) ? end1 : (mask & 1) ? end0 : 0;      // (only up to the end of the last active segment)
    A.mask = mask;
    for (int s = 0; s < 3; ++s) {
        const int t = steps[s] > 0 ? steps[s] : 1;
        A.bc1[s] = (float)(1.0 - pow((double)beta1, (double)t));
        A.bc2s[s] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    }
    if (A.end[2] <= 0) return 0;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(cdiv(A.end[2], 256)), dim3(256), 0, S_(st), param, grad, m, v, A, lr, beta1, beta2, eps, grad_scale);
    CHECK_LAUNCH();
    return 0;
}

size_t lush_step_state_bytes(void) { return sizeof(StepState); }
int lush_step_state_init(void* state, unsigned long long draw_base, int global_step, const int* adam_steps, double lrate,
                         double decay_steps, double beta1, double beta2, lush_stream_t st) {
    if (!state || !adam_steps) return set_error("lush_step_state_init: null argument");
    hipLaunchKernelGGL(step_state_init_kernel, dim3(1), dim3(1), 0, S_(st), static_cast<StepState*>(state), draw_base, global_step,
                       adam_steps[0], adam_steps[1], adam_steps[2], lrate, decay_steps, beta1, beta2);
    CHECK_LAUNCH();
    return 0;
}
int lush_step_state_advance(void* state, int n_draw_calls, int active_mask, double lrate, double decay_steps, double beta1,
                            double beta2, lush_stream_t st) {
    if (!state) return set_error("lush_step_state_advance: null state");
    hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(1), 0, S_(st), static_cast<StepState*>(state), n_draw_calls,
                       active_mask, lrate, decay_steps, beta1, beta2);
    CHECK_LAUNCH();
    return 0;
}
int lush_adam_state(float* param, const float* grad, float* m, float* v, long long n, const void* state, int segment,
                    float beta1, float beta2, float eps, float grad_scale, lush_stream_t st) {
    if (n <= 0) return 0;
    if (!state || segment < 0 || segment > 2) return set_error("lush_adam_state: bad state / segment");
    hipLaunchKernelGGL(adam_state_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(st), param, grad, m, v, n,
                       static_cast<const StepState*>(state), segment, beta1, beta2, eps, grad_scale);
    CHECK_LAUNCH();
    return 0;
}

int lush_adam_state_multi(float* param, const float* grad, float* m, float* v, long long end0, long long end1, long long end2,
                          const void* state, int mask, float beta1, float beta2, float eps, float grad_scale, lush_stream_t st) {
    if (!state || end0 < 0 || end1 < end0 || end2 < end1 || mask < 0 || mask > 7) return set_error("lush_adam_state_multi: bad state / segment ends / mask");
    // (only up to the end of the last active segment)
    const long long n = (mask & 4) ? end2 : (mask & 2) ? end1 : (mask & 1) ? end0 : 0;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_state_multi_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(st), param, grad, m, v, end0, end1, n,
                       static_cast<const StepState*>(state), mask, beta1, beta2, eps, grad_scale);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
