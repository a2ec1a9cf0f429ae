// lush-march: layout of the packed MLP weights and the argument blocks of the
// fused MLP kernels.  Shared by host launchers and device code.
//
// One MLP (NeRF: HW=256, NL=8, SKIP=5; NeRF_Noise: HW=128, NL=4, SKIP=-1;
// utils/run_lushnerf_helpers.py:365-423, 456-512) is a list of GEMM *segments*
//     out[rows][pts] += Wseg[rows][K] * img[K][pts]
// Each segment is stored as MFMA A-operand fragments, NS bf16 planes each:
//     entry (kk, rb): [plane][lane 0..63][8 x bf16]   (NS KiB)
//     lane (r = lane&31, h = lane>>5) element j = Wseg[32*rb + r][16*kk + 8*h + j]
// so a wave fetches one fragment with one fully coalesced 1-KiB load.
#pragma once
#include <stdint.h>

namespace lush {

enum { NET_MAX_LAYERS = 8 };
enum { PLANES_F16 = 17 };   // plane code: 1..3 = bf16 planes, 17 = ONE fp16 plane (forward of the NeRF nets)
// kernel-variant bits of the C ABI (include/lush_march.h LUSH_VARIANT_*): 0 = the product's choice
enum { LUSH_VARIANT_FWD_HALF = 1, LUSH_VARIANT_FWD_512 = 2, LUSH_VARIANT_BWD_512 = 4, LUSH_VARIANT_HEAD_KERNEL = 8, LUSH_VARIANT_BWD_HALF = 16,
       LUSH_VARIANT_PE_ROWS = 64, LUSH_VARIANT_DW_SPLIT = 128, LUSH_VARIANT_DENSE_BWD = 256, LUSH_VARIANT_KERNEL_BITS = 0x3DF };

template <int HW_, int NL_, int SKIP_>
struct NetT {
    static constexpr int HW = HW_, NL = NL_, SKIP = SKIP_;
    static constexpr int HV = HW / 2;
    static constexpr int NRB = HW / 32, NRBV = HV / 32;
    static constexpr int KKH = HW / 16, KKV = HV / 16;
    static constexpr int KKX = 4, KKD = 2;   // gamma(x) padded to 64, gamma(d) to 32

    // ---- forward segments, offsets in entries ----
    // L0 | L1.. (skip layer = [a: gamma(x) part][b: h part]) | FEAT | ALPHA | VA | VB | RGB
    static constexpr int fwd_L(int l, bool part_b) {
        int off = 0;
        if (l == 0) return 0;
        off += KKX * NRB;
        for (int i = 1; i < l; ++i) off += (i == SKIP ? (KKX + KKH) : KKH) * NRB;
        if (l == SKIP && part_b) off += KKX * NRB;
        return off;
    }
    static constexpr int fwd_FEAT = fwd_L(NL, false);
    static constexpr int fwd_ALPHA = fwd_FEAT + KKH * NRB;
    static constexpr int fwd_VA = fwd_ALPHA + KKH;
    static constexpr int fwd_VB = fwd_VA + KKH * NRBV;
    static constexpr int fwd_RGB = fwd_VB + KKD * NRBV;
    static constexpr int fwd_END = fwd_RGB + KKV;

    // ---- backward (transposed) segments ----
    // VAT (rows HW, K=HV) | VBT (rows 32, K=HV) | FEATT | L{NL-1}T .. L1T | L0T (rows 64)
    //   skip layer T = [a: rows 64 (gamma(x) part)][b: rows HW]
    static constexpr int KVB = 2 * NRB;      // VBT is packed with its K zero-padded to this many k-blocks (whole chain-kernel positions)
    static_assert(KVB >= KKV && KVB <= KKH, "VBT padding");
    static constexpr int bwd_VAT = fwd_END;
    static constexpr int bwd_VBT = bwd_VAT + KKV * NRB;
    static constexpr int bwd_FEATT = bwd_VBT + KVB;
    static constexpr int bwd_LT(int l, bool part_b) {   // l in [0, NL-1]
        int off = bwd_FEATT + KKH * NRB;
        for (int i = NL - 1; i > l; --i) off += (i == SKIP ? (KKH * 2 + KKH * NRB) : KKH * NRB);
        if (l == SKIP && part_b) off += KKH * 2;
        return off;
    }
    static constexpr int bwd_END = bwd_LT(0, false) + KKH * 2;
    // ---- second copy of the forward segments for the register-resident chain kernel
    // (lush_mlp_chain.hip): same order and offsets, output rows permuted inside each 32-row block
    // (chain_row()) so that a lane's 16 accumulators are the 2 x 8 consecutive features it supplies
    // as the next layer's B operand ----
    static constexpr int fwd2_base = bwd_END;
    // ... and of the backward (transposed) segments for mlp_chain_bwd_kernel: bwd2_X = bwd2_base + (bwd_X - bwd_VAT)
    static constexpr int bwd2_base = fwd2_base + fwd_END;
    // third copy of the forward segments for the 256-register chain kernel (mlp_chain_fwd_half_kernel): rows permuted by
    // chain_row() as in the second copy, but every full-width segment is cut into its two 128-row halves and the stream
    // runs [half 0 of the layer | half 1 of the layer] (skip layer: [a0 | b0 | a1 | b1]); narrow segments unchanged.
    static constexpr int fwd3_base = bwd2_base + (bwd_END - bwd_VAT);
    // ... and of the transposed segments for mlp_chain_bwd_half_kernel: [VAT h0 | VAT h1 | VBT | FEATT h0 | h1 |
    // per layer NL-1..1: (skip: a) b h0 | b h1 | L0T], every full-width segment in 128-row halves
    static constexpr int bwd3_base = fwd3_base + fwd_END;
    // fourth copy of the forward segments for the 64-points-per-wave kernel (mlp_wide_fwd_kernel, lush_mlp_wide.hip):
    // rows permuted by chain_row(); every full-width segment cut into 64-row QUARTERS, the stream runs quarter by quarter
    // (skip layer: [x part | h part] per quarter) in positions of 8 fragments = 4 k-blocks x 2 row blocks (k-block major);
    // ALPHA 2 positions of 8 k-blocks; the views layer per 64-row quarter [gamma(d) part, K zero-padded to 4 k-blocks |
    // feature part]; RGB one position of 8 k-blocks.
    static constexpr int fwd4_base = bwd3_base + (bwd_END - bwd_VAT);
    static constexpr int fwd4_len = fwd_ALPHA + KKH + (HV / 64) * (8 + 2 * KKH) + KKV;
    // ... and of the transposed segments for mlp_wide_bwd_kernel (lush_mlp_wide_bwd.hip), in execution order: VAT per 64-row
    // quarter (K = HV: 2 positions) | VBT (one position of 8 k-blocks x 1 row block) | FEATT per quarter (4 positions) |
    // per layer NL-1..1: its four quarters, then (skip layer) the 64 gamma(x) rows as one more quarter | L0T (64 rows).
    static constexpr int bwd4_base = fwd4_base + fwd4_len;
    static constexpr int bwd4_len = (HW / 64) * 2 * KKV + KKV + (HW / 64) * 2 * KKH * NL + (SKIP > 0 ? 2 * KKH : 0) + 2 * KKH;
    static constexpr int total_entries = bwd4_base + bwd4_len;

    static constexpr int n_mask_layers = NL + 1;   // h_0..h_{NL-1}, hv

    // ---- fp32 block appended to the packed fragments (offsets in floats) ----
    // biases of every layer and the two K<=3 head matrices the backward applies on the VALU
    static constexpr int f32_b_trunk = 0;                    // [NL][HW]
    static constexpr int f32_b_feat = NL * HW;               // [HW]
    static constexpr int f32_b_alpha = f32_b_feat + HW;      // [1] (32 reserved)
    static constexpr int f32_b_views = f32_b_alpha + 32;     // [HV]
    static constexpr int f32_b_rgb = f32_b_views + HV;       // [3] (32 reserved)
    static constexpr int f32_w_rgb = f32_b_rgb + 32;         // [3][HV]
    static constexpr int f32_w_alpha = f32_w_rgb + 3 * HV;   // [HW]
    static constexpr int f32_total = f32_w_alpha + HW;
};

typedef NetT<256, 8, 5> NetNerf;
typedef NetT<128, 4, -1> NetNoise;

// One pack job: fill `n_entries` = KK*NRB fragments of a segment from an fp32
// matrix.  Element (row, k) of the segment = src[row*sr + k*sk] if row < rows
// and k < cols, else 0.
// MFMA row rho (0..31) of a permuted block holds feature chain_row(rho): the lane (col, h) that owns
// rows (q&3) + 8*(q>>2) + 4h, q = 0..15, then owns features 16*(q>>3) + 8h + (q&7).
__host__ __device__ constexpr int chain_row(int rho) {
    const int h = (rho >> 2) & 1, q = (rho & 3) + 4 * (rho >> 3);
    return 16 * (q >> 3) + 8 * h + (q & 7);
}

struct PackJob {
    const float* src;
    int sr, sk;
    int rows, cols;
    int nrb, kk;
    int perm;            // 1: rows of each 32-row block permuted by chain_row()
    int dst_entry;       // first entry in the packed buffer
    int first_block;     // prefix sum of (nrb*kk) over previous jobs
};

struct PackTable {
    PackJob j[64];
    int n;
};

// A pack PLAN: every fragment segment and fp32 block of SEVERAL networks / plane codes as one device-resident table, so that
// a training step re-packs all its weights with ONE launch (pack_plan_kernel) instead of three per network and direction.
struct PlanJob {
    PackJob j;           // first_block = prefix sum over the whole plan; dst_entry relative to `dst`
    void* dst;           // the packed buffer this job writes into
    int code;            // plane code of that buffer (1..3 bf16 planes, PLANES_F16)
    int kind;            // 0 = fragments; 1 / 2 = fp32 block of a NetNerf / NetNoise (64 elements per block), params = prm[prm_index]
    int prm_index;
    int pad;
};
constexpr int PLAN_MAX_NETS = 8;
struct PlanHeader {
    int n_jobs, total_blocks, n_prm, pad;
    // followed by MlpParams prm[PLAN_MAX_NETS], then PlanJob jobs[n_jobs]
};

struct MlpParams {       // device pointers to the fp32 nn.Linear parameters
    const float* w[NET_MAX_LAYERS];
    const float* b[NET_MAX_LAYERS];
    const float *w_feat, *b_feat, *w_alpha, *b_alpha, *w_views, *b_views, *w_rgb, *b_rgb;
};

struct MlpGrads {        // same shapes as MlpParams, fp32, accumulated with atomics
    float* w[NET_MAX_LAYERS];
    float* b[NET_MAX_LAYERS];
    float *w_feat, *b_feat, *w_alpha, *b_alpha, *w_views, *b_views, *w_rgb, *b_rgb;
};

// Activation stash written by the forward kernel (read by backward / dW).  Every array is
// [plane][Ppad][cols] bf16; the per-layer arrays are equally spaced (h[l] = h0 + l*h_stride) so the
// kernels carry a handful of scalars instead of a pointer table (a 60-pointer argument block spills
// SGPRs and puts scratch reloads into the MFMA loops).
struct MlpFwdArgs {
    const float* rays;             // [R][11]
    const float* z;                // [R][S]
    int S, P, n_tiles;
    const uint4* wpk;              // packed fragments (NS planes) followed by the fp32 block
    float* raw;                    // [P][4]
    unsigned long long* mask;      // ReLU sign bits, see mask_index()
    void* mask_dummy;              // 4 KiB nobody reads: where a kernel that runs the feature layer through its ReLU-layer code sends the words
    __bf16* pe;                    // [NS][Ppad][PE_ROW]
    float* xd;                     // [Ppad][8]: x (3), 0, viewdir (3), 0 -- what the weight gradients re-encode when pe_rows == 0
    int pe_rows;                   // write the encoded rows to `pe` (kernels / variants whose weight gradients read them)
    __bf16* h0;                    // h_l = h0 + l*h_stride, each [sp][Ppad][HW]
    __bf16* feat;                  // [sp][Ppad][HW]
    __bf16* hv;                    // [sp][Ppad][HV]
    long long h_stride;            // elements between consecutive layers' arrays
    long long plane_pe, plane_h, plane_hv;   // plane strides in elements
    int write_stash;
    int stash_planes;              // planes copied to the stash (<= NS): what the backward will use
    // Live-point launches (round 5; the 64-points-per-wave and the 128-point chain kernels of the 8x256 net): point i of the launch is point live_idx[i] of the
    // [R][S] grid, the point count is read from *live_cnt on the device (P / n_tiles then are upper bounds: the grid's size),
    // `raw` is not written (the caller has it from the pass over all points).  Both null: a launch over all the points.
    const int* live_idx;
    const int* live_cnt;
};

struct MlpBwdArgs {
    const float* rays;
    const float* z;
    int S, P, n_tiles;
    const uint4* wpk;              // packed fragments with NSB planes + fp32 block
    const float* draw;             // [P][4]
    const unsigned long long* mask;
    __bf16* dz0;                   // dZ_l = dz0 + l*dz_stride, each [NSB][Ppad][HW]
    __bf16* dfeat;                 // [NSB][Ppad][HW]
    __bf16* dzv;                   // [NSB][Ppad][HV]
    long long dz_stride;
    long long plane_h, plane_hv;
    float* dpts;                   // [P][8]: d/dx (3), pad, d/dviewdir (3), pad
    const float* scale;            // fp16 chain only: {loss scale, 1/scale} written by grad_scale_kernel
    const int* live_idx;           // live-point launches (MlpFwdArgs): draw, mask, dZ, dpts are indexed by the launch's point number
    const int* live_cnt;
};

struct DwArgs {
    const __bf16* Z; long long z_plane; int ldz; int n_out;   // dZ [Ppad][ldz]
    const __bf16* X; long long x_plane; int ldx; int xcol0; int k_in;
    float* dW; int ldw; int wcol0;
    float* db;                     // may be null
    int x_f16;                     // X stash holds fp16 (one plane) instead of bf16 planes
    int z_f16;                     // dZ holds loss-scaled fp16 (one plane): fp16 MFMA, epilogue multiplies by scale[1]
    const float* scale;            // {loss scale, 1/scale} of the fp16 gradient chain, or null
    int Ppad;                      // multiple of 32
    int pts_per_split;             // multiple of 32
};

// One weight-gradient GEMM of a grouped launch (dw_group_kernel): dW[o][wcol0 + i] += sum_p Z[p][o] X[p][xcol0 + i],
// optionally a second input block that shares the same Z (the skip layer's gamma(x) columns, the views layer's
// gamma(d) columns): dW[o][wcol2 + i] += sum_p Z[p][o] X2[p][x2col0 + i], and the bias gradient db[o] += sum_p Z[p][o].
struct DwJob {
    const __bf16* Z; int ldz; int n_out;
    const __bf16* X; int ldx; int xcol0; int k_in;
    const __bf16* X2; int ldx2; int x2col0; int k2_in;     // X2 == null: none; k2_in <= 64
    long long z_plane, x_plane, x2_plane;                   // plane strides in elements (two-plane launches)
    float* dW; int ldw; int wcol0;
    float* dW2; int ldw2; int wcol2; int n_out2;            // where the X2 columns go (usually dW / ldw again), rows < n_out2
    int pe_mode;                                            // DwGroup::xd given: X2 is computed, not read: 1 = gamma(x), 2 = gamma(viewdir)
    float* db;                                              // may be null
    int pps;                                                // DwGroup::per_job == 2: points per slice of THIS job (multiple of 32)
};
enum { DW_MAX_JOBS = 12 };
// Passes of at most LUSH_DW_PERJOB_MAX_PTS points run ONE job per workgroup with slices of at least LUSH_DW_PERJOB_MIN_PTS points
// (4 096 points, 11 jobs: 256 -> 70 us, 512 -> 64 us, 1024 -> 88 us); see lush_abi.hip (host's choice) and dw_group_kernel (live launches).
#ifndef LUSH_DW_PERJOB_MAX_PTS
#define LUSH_DW_PERJOB_MAX_PTS 262144
#endif
#ifndef LUSH_DW_PERJOB_MIN_PTS
#define LUSH_DW_PERJOB_MIN_PTS 512
#endif
// A one-plane dZv row carries 8 more columns: the head gradients [d_r d_g d_b d_alpha] as a hi and a lo 16-bit plane
// (hi = round16(x), lo = round16(x - hi): 16 / 22 bits for bf16 / fp16), scaled like dZ.  The grouped weight-gradient
// launch then gets the K<=3 heads as 8 more GEMM rows of the feature job (row HV+3 + row HV+7 = d_alpha^T h_{NL-1}) and
// one small job (Z = these 8 columns, X = views hidden): no kernel re-reads h_{NL-1} for the alpha head.
constexpr int DZV_EXT = 8;

// The feature layer has no activation, so with G = dZv^T h_{NL-1} [HV][HW] and s = sum dZv (what the grouped launch
// accumulates for the one-plane backward):  dW_feat = Wva^T G,  db_feat = Wva^T s,  dW_views[:, :HW] = G Wf^T + s b_f^T,
// db_views = s  -- neither the feature activations nor their gradients travel through HBM.
struct FeatFactorArgs {
    const float* G; const float* s;               // scratch of this launch (fp32, already unscaled); heads: DZV_EXT more rows
    const float* Hd; const float* sH;             // heads folded in (or null): [DZV_EXT][HV] = (d_rgb, d_alpha planes)^T hv, and its sums
    float* g_w_alpha; float* g_b_alpha;           // may be null (the noise net's alpha head has no gradient)
    float* g_w_rgb; float* g_b_rgb;
    const float* w_views; const float* w_feat; const float* b_feat;
    float* g_w_feat; float* g_b_feat; float* g_w_views; float* g_b_views;   // accumulated into
    int HW, HV, ldv;                              // ldv = row length of w_views (HW + gamma(d) columns)
};
struct DwGroup {
    DwJob j[DW_MAX_JOBS];
    int n;
    int Ppad;                      // multiple of 32
    int pts_per_split;             // multiple of 32
    const float* scale;            // {loss scale, 1/scale} when Z is loss-scaled fp16, else null
    const float* xd;               // [Ppad][8] points and view directions (MlpFwdArgs::xd), or null: every X2 is read from its rows
    int per_job;                   // 1: grid (splits, n) -- a workgroup takes ONE job of its slice (launches whose slices alone leave
                                   // CUs idle); 0: grid (splits) -- a workgroup takes every job of its slice in turn;
                                   // 2: grid (sum of the jobs' slice counts) -- ONE job per workgroup on a slice of DwJob::pps points;
                                   // workgroup b belongs to the job j with first[j] <= b < first[j + 1].  The slice counts are
                                   // proportional to the jobs' cost per point, so every workgroup is busy for the same time and
                                   // drains / flushes / refills ONCE (a walking workgroup does it per job: ten times)
    int first[DW_MAX_JOBS + 1];    // per_job == 2: prefix sums of the jobs' slice counts
    const int* live_cnt;           // live-point launches: the point count on the device (Ppad / pts_per_split are derived from it
                                   // in the kernel; the host's values are upper bounds), or null
};

}  // namespace lush
