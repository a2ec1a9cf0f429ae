// lush-march: shared device helpers (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lush {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;   // 8 bytes
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator block
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Diagnostic build -DLUSH_CLOCK (tools/clock_probe.py; never the product): the clock a kernel actually runs at, per workgroup, as
// d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Thread 0 of a workgroup stamps both
// counters at the kernel's head and at its end into an array of its own that no other code reads; no output value depends on them.
#ifdef LUSH_CLOCK
#define LUSH_CLOCK_DECL(name) __device__ unsigned long long name[1024][4];
#define LUSH_CLOCK_STAMP(name, k)                                                                      \
    do {                                                                                               \
        const unsigned lc_b = blockIdx.x + blockIdx.y * gridDim.x;                                     \
        if (threadIdx.x == 0 && lc_b < 1024) {                                                         \
            name[lc_b][2 * (k)] = __builtin_amdgcn_s_memtime();                                        \
            name[lc_b][2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();                                \
        }                                                                                              \
    } while (0)
#define LUSH_CLOCK_EXPORT(fn, name)                                                                    \
    extern "C" int fn(unsigned long long* out /* host, 1024 x 4; zeroed on the device afterwards */) { \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                           \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(name), sizeof(unsigned long long) * 4096) != hipSuccess) return -1; \
        static unsigned long long zeros[4096];                                                         \
        return hipMemcpyToSymbol(HIP_SYMBOL(name), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;       \
    }
#else
#define LUSH_CLOCK_DECL(name)
#define LUSH_CLOCK_STAMP(name, k)
#define LUSH_CLOCK_EXPORT(fn, name)
#endif

enum { DT_BF16 = 0, DT_F16 = 1 };   // 16-bit operand element type (see mfma_f16 below)
constexpr int WAVE = 64;
constexpr int MAX_PLANES = 3;

// Positional-encoding image: one 256-byte row (128 bf16) per point and plane.
//   cols 0..62   gamma(x)  (x, then per frequency sin xyz, cos xyz), col 63 = 0
//   cols 64..90  gamma(d)  (same order, 4 frequencies),             cols 91..95 = 0
//   cols 96..127 never read (keeps the XOR swizzle inside one 256-B row)
constexpr int PE_X = 64;
constexpr int PE_D = 32;
constexpr int PE_ROW = 128;
constexpr int PE_X_VALID = 63;
constexpr int PE_D_VALID = 27;
constexpr int L_X = 10;   // multires        (utils/run_lushnerf_helpers.py:347-361)
constexpr int L_D = 4;    // multires_views

// Row of a 32x32 MFMA accumulator held in register q of a lane in half h
// (cdna_hip_programming.md section 3: row=(reg&3)+8*(reg>>2)+4*(lane>>5)).
__device__ __forceinline__ constexpr int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// Byte offset of 16-byte chunk `chunk` of row `row` in an XOR-swizzled LDS image
// whose rows are a multiple of 256 bytes (T2 swizzle: conflict-free ds_read_b128
// when the 32 lanes of a half-wave read 32 different rows at the same chunk).
__device__ __forceinline__ int swz(int row, int chunk, int row_bytes) {
    return row * row_bytes + ((chunk ^ (row & 15)) << 4);
}

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }

// Split x into NS bf16 planes: x ~= p0 + p1 (+ p2), each the round-to-nearest
// bf16 of the running remainder.  Two planes carry 16 mantissa bits, three 24.
template <int NS, int DT = DT_BF16>
__device__ __forceinline__ void split_planes(float x, __bf16 (&p)[NS]) {
    if constexpr (DT == DT_F16) {
        static_assert(NS == 1, "fp16 is a single-plane format here");
        p[0] = __builtin_bit_cast(__bf16, (_Float16)x);
    } else {
        float r = x;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = (__bf16)r;
            if (i + 1 < NS) r = r - (float)p[i];
        }
    }
}
// Two values at once, packed per plane (element 0 in the low half): lets the compiler use both operands
// of v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 instead of converting singly and merging with v_perm.
template <int NS, int DT = DT_BF16>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&o)[NS]) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    f32x2_t r = {a, b};
    if constexpr (DT == DT_F16) {
        static_assert(NS == 1, "fp16 is a single-plane format here");
        o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const bf16x2_t p = __builtin_convertvector(r, bf16x2_t);
            o[i] = __builtin_bit_cast(unsigned, p);
            if (i + 1 < NS) r = r - __builtin_convertvector(p, f32x2_t);
        }
    }
}
// value of a stored 16-bit element
template <int DT>
__device__ __forceinline__ float elem_to_f32(__bf16 v) {
    if constexpr (DT == DT_F16) return (float)__builtin_bit_cast(_Float16, v);
    else return (float)v;
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16-bit operand element type of a kernel: DT_BF16 planes (1..3 of them) or ONE fp16 plane.
// fp16 (11-bit mantissa) in a single plane keeps render outputs within ~2.5e-5 of fp32 -- inside the
// 1e-4 bound at one MFMA per product -- but its range is unsafe for gradients, so only the forward of
// the 8x256 nets uses it.  Storage stays typed __bf16 (a 16-bit container); fp16 values are bit-cast.
__device__ __forceinline__ f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// acc += sum over plane pairs of A_i * B_j keeping every product of order
// <= 2^-8(NS-1): NS=1 -> 1 MFMA, NS=2 -> 3, NS=3 -> 6.  Small terms first.
template <int NS, int DT = DT_BF16>
__device__ __forceinline__ f32x16 mfma_planes(const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x16 c) {
    if constexpr (DT == DT_F16) return mfma_f16(a[0], b[0], c);
    if constexpr (NS == 3) {
        c = mfma_bf16(a[1], b[1], c);
        c = mfma_bf16(a[2], b[0], c);
        c = mfma_bf16(a[0], b[2], c);
    }
    if constexpr (NS >= 2) {
        c = mfma_bf16(a[1], b[0], c);
        c = mfma_bf16(a[0], b[1], c);
    }
    c = mfma_bf16(a[0], b[0], c);
    return c;
}

__host__ __device__ constexpr int mfma_per_product(int ns) { return ns == 1 ? 1 : (ns == 2 ? 3 : 6); }

// wave64 reductions / scans -------------------------------------------------
// By DPP (data-parallel primitives on the VALU: the other lane's value is an operand modifier), the sequence the compiler's own
// wave scans use on gfx9: an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals carried
// across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Lanes without a source take the identity.
// As __shfl_xor / __shfl_up (ds_bpermute through the LDS crossbar, a wait and an add per step) the nine sums of
// ray_grad_reduce_kernel were 54 LDS round trips per ray -- most of its 20 us; a DPP step is one VALU instruction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wave_dpp(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_sum(float v, int /*lane*/ = 0) {
    v += wave_dpp<0x111, 0xf>(0.f, v);        // row_shr:1
    v += wave_dpp<0x112, 0xf>(0.f, v);        // row_shr:2
    v += wave_dpp<0x114, 0xf>(0.f, v);        // row_shr:4
    v += wave_dpp<0x118, 0xf>(0.f, v);        // row_shr:8
    v += wave_dpp<0x142, 0xa>(0.f, v);        // row_bcast:15 -> rows 1, 3
    v += wave_dpp<0x143, 0xc>(0.f, v);        // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float wave_incl_prod(float v, int /*lane*/ = 0) {
    v *= wave_dpp<0x111, 0xf>(1.f, v);
    v *= wave_dpp<0x112, 0xf>(1.f, v);
    v *= wave_dpp<0x114, 0xf>(1.f, v);
    v *= wave_dpp<0x118, 0xf>(1.f, v);
    v *= wave_dpp<0x142, 0xa>(1.f, v);
    v *= wave_dpp<0x143, 0xc>(1.f, v);
    return v;
}
// the sum over the wave, in every lane (lane 63 of the inclusive scan, read into a scalar)
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_incl_sum(v)), 63));
}

}  // namespace lush
