/* lush_march.h -- C ABI of liblush_march.so: the MI355X (gfx950) ray-march hot
 * path of LuSh-NeRF.
 *
 * The reference has no FFI; its boundary for this path is a set of Python
 * methods on an nn.Module.  Each entry point below names
 * the reference code it replaces (file:line into the LuSh-NeRF tree).  The
 * Python mirror in lush_nerf_amd/ binds these with ctypes and wraps them in
 * torch.autograd.Function objects that keep the reference signatures.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless marked host;
 *  - all work is enqueued on `stream` (a hipStream_t); no call synchronises,
 *    allocates device memory, or keeps state between calls;
 *  - return value 0 = ok, negative = error; lush_last_error() gives a
 *    thread-local message;
 *  - tensors are fp32, row-major, contiguous, with the shapes shown;
 *  - "accumulate" outputs are added to, everything else is overwritten.
 */
#ifndef LUSH_MARCH_H
#define LUSH_MARCH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lush_stream_t;

const char* lush_last_error(void);
int lush_abi_version(void);   /* 10 (round 5; 9: lush_rbk_mlp_bwd consumes its d_rvw rows; 10: the live-point entry points lush_live_compact /
                             * lush_mlp_fwd_live / lush_mlp_bwd_*_live / lush_ray_grad_reduce_live, LUSH_VARIANT_DENSE_BWD,
                             * LUSH_VIEW_LIVE_COUNTS; the march's one- and two-plane backward runs on the live points).  Earlier: 8 (round 3: explicit kernel-variant argument instead of environment switches; 6: the blur-mix / tone-map
                             * backward entry points write their outputs instead of accumulating; 7, round 4: the fused ray-level
                             * entry points lush_rbk_warp_ndc_* and lush_blur_mix_*, an explicit d_rvw row stride, and no second
                             * stream inside lush_march_bwd; 8: lush_pack_plan_run takes a buffer to clear, lush_adam_multi /
                             * lush_adam_state_multi update several segments in one launch) */

/* ------------------------------------------------------------------ sampling
 * z grid + stratified jitter: models/lushnerf.py:389-412 / 501-523.
 * rays [R][11] = [o(3) d(3) near far viewdir(3)]; t_rand [R][S] or NULL
 * (perturb == 0); z [R][S]. */
int lush_zgrid(const float* rays, int R, int S, int lindisp, const float* t_rand, float* z,
               lush_stream_t stream);
/* z of sample `index` of the un-jittered grid (the noise MLP's point,
 * models/lushnerf.py:271, 396, 612): z [R]. */
int lush_zfixed(const float* rays, int R, int S, int index, int lindisp, float* z, lush_stream_t stream);

/* ------------------------------------------------------- numerical-fault word
 * The reference tests every entry of render_rays' result dict for NaN/Inf after every chunk and
 * prints (models/lushnerf.py:474-478, 578-582): one device sync per key per chunk.  Here the
 * kernels that produce those entries OR bits into ONE caller-owned int32 word (`flags`, may be
 * NULL = no checking), which the caller reads whenever it likes.  lush_composite_fwd shifts
 * its five bits left by `flag_shift` (0 for the final pass: rgb_map, depth_map, acc_map,
 * density_map, raw; LUSH_FAULT_COARSE_SHIFT for the coarse pass: rgb0, depth0, acc0, density0). */
#define LUSH_FAULT_RGB 1
#define LUSH_FAULT_DEPTH 2
#define LUSH_FAULT_ACC 4
#define LUSH_FAULT_DENSITY 8
#define LUSH_FAULT_RAW 16
#define LUSH_FAULT_COARSE_SHIFT 5
#define LUSH_FAULT_ZSTD 1024

/* ---------------------------------------------------------------- compositing
 * NeRFAll.raw2outputs, models/lushnerf.py:296-352.  raw [R][S][4]; noise [R][S-1]
 * N(0,1) draws or NULL; near_mask < 0 disables the eval-only near-plane mask
 * (:331-335), otherwise density is zeroed where z[j+1] <= near_mask.
 * Outputs rgb [R][3], depth [R], acc [R], weights [R][S], density [R][S-1]. */
int lush_composite_fwd(const float* raw, const float* z, const float* rays, int R, int S,
                       const float* noise, float noise_std, float near_mask, int white_bkgd,
                       float* rgb, float* depth, float* acc, float* weights, float* density,
                       int* flags, int flag_shift, lush_stream_t stream);
/* Backward of the above w.r.t. raw and rays_d.  g_* may be NULL (= 0).
 * draw [R][S][4] overwritten; drays [R][11]: columns 3..5 accumulate, or -- init_drays != 0 -- the whole row is written
 * (zeros elsewhere: the first pass of a march needs no zero-fill of its own).
 * Folded in so that they cost no launch of their own (any of them may be NULL / 0):
 *   block_max  [lush_composite_bwd_blocks(R)] floats: per-workgroup max |d_raw|, from which lush_loss_scale (one small launch)
 *              makes the loss scale of the fp16 gradient chain -- instead of a pass of its own over d_raw;
 *   zero_buf / zero_n: floats zeroed by this launch (the scratch a later launch accumulates into). */
int lush_composite_bwd_blocks(int R);
int lush_composite_bwd(const float* raw, const float* z, const float* rays, int R, int S,
                       const float* noise, float noise_std, float near_mask, int white_bkgd,
                       const float* g_rgb, const float* g_depth, const float* g_acc,
                       float* draw, float* drays, float* block_max, float* zero_buf, long long zero_n, int init_drays,
                       lush_stream_t stream);
/* scale2 = {s, 1/s}: s = the power of two that puts max(block_max[0..n)) into [8, 16); 1 when the maximum is 0 or not finite
 * (the rule of the loss-scaled fp16 gradient chain). */
int lush_loss_scale(const float* block_max, int n, float* scale2, lush_stream_t stream);

/* ------------------------------------------------------ hierarchical sampling
 * sample_pdf (utils/run_lushnerf_helpers.py:566-609) on bins = mid-points and
 * weights[1:-1], followed by sort(cat(z, z_samples)) (models/lushnerf.py:435-440,
 * 544-549).  u [R][Ni] U[0,1) draws, or NULL for the deterministic linspace.
 * z_out [R][S+Ni] sorted; z_samples [R][Ni] (may be NULL); z_std [R] =
 * std(z_samples, unbiased=False) (:465).
 * LIMITS: 3 <= S <= 256 and S + Ni <= 512 (one wavefront per ray: the CDF and the merged sort live in fixed LDS
 * arrays); anything larger is REFUSED with an error, nothing is launched.  BASELINE's largest config is 128 + 128. */
int lush_sample_merge(const float* z, const float* weights, int R, int S, int Ni, const float* u,
                      float* z_out, float* z_samples, float* z_std, int* flags, lush_stream_t stream);

/* ------------------------------------------------------------- ray prologue
 * The shared head of render_infer / render_train_scene / render_train_noise
 * (models/lushnerf.py:706-729, 772-795, 827-850) with ndc_rays
 * (utils/run_lushnerf_helpers.py:542-562): rays [N][3][2] -> batch [N][11].
 * cx = -1/(W/(2 focal)), cy = -1/(H/(2 focal)) rounded to fp32 by the caller. */
int lush_pack_rays_fwd(const float* rays, int N, int ndc, float cx, float cy, float near, float far,
                       float* batch, lush_stream_t stream);
/* dbatch [N][11] -> drays [N][3][2] (overwritten). */
int lush_pack_rays_bwd(const float* rays, int N, int ndc, float cx, float cy, const float* dbatch,
                       float* drays, lush_stream_t stream);

/* Device-side ray table: get_rays / get_rays_np
 * (utils/run_lushnerf_helpers.py:517-539) for N (view, pixel) pairs instead of the pre-materialised
 * [N_img*H*W, 2, 3] table and its host permutation (run_lushnerf.py:561-589, 610-614).
 * c2w [V][3][4]; view, px, py [N] int64; rays [N][3][2]. */
int lush_gen_rays(const float* c2w, const int64_t* view, const int64_t* px, const int64_t* py, int N,
                  float fx, float fy, float cx, float cy, float* rays, lush_stream_t stream);

/* The same for all H*W pixels of one pose, row-major pixel order: the eval path's get_rays call
 * (models/lushnerf.py:868-896 -> helpers:517-528).  c2w [3][4]; rays [H*W][3][2]. */
int lush_gen_rays_image(const float* c2w, int H, int W, float fx, float fy, float cx, float cy,
                        float* rays, lush_stream_t stream);

/* ------------------------------------------------ consistency branch
 * Ray gather of NeRFAll.Render_Aligned_Pixel (models/lushnerf.py:949-985): for pose v and sample s
 * the matched pixel (x, y) = align[v][samples[s]][2:4].long() clamped to the image, its ray taken
 * from get_rays(H, W, K, c2w[v]).  c2w [V][3][4]; align [V][HW][4] = Align_matrix[anchor];
 * cert [V][HW] = Align_mask[anchor] as fp32 (cert_is_u8 = 0) or bool/uint8 (1; the reference
 * allocates it as torch.bool, run_lushnerf.py:292); samples [ns] int64.
 * rays [V*ns][3][2]; cert_out [V][ns] fp32. */
int lush_align_rays(const float* c2w, const float* align, const void* cert, int cert_is_u8,
                    const int64_t* samples, int V, int ns, long long HW, int H, int W, float fx, float fy,
                    float cx, float cy, float* rays, float* cert_out, lush_stream_t stream);
/* compute_mean_with_confidence (utils/run_lushnerf_helpers.py:665-688) and the masked L1 of
 * run_lushnerf.py:644-650: loss[0] = sum |rgb - mean| * (cert >= threshold) / #(cert >= threshold)
 * (overwritten; NaN when nothing passes the threshold, as in the reference), grad [V][ns][3] =
 * d loss / d rgb including the path through the mean.  rgb [V][ns][3], cert [V][ns]. */
int lush_consist_loss_fwd_bwd(const float* rgb, const float* cert, int V, int ns, float threshold,
                              float* loss, float* grad, lush_stream_t stream);

/* ----------------------------------------------------------- blur kernel (RBK)
 * View_Embedding + Rigid_Blurring_Kernel.forward trunk/heads,
 * models/lushnerf.py:27-35, 118-148.  The MLP input is the image embedding only,
 * so it is evaluated once per IMAGE (num_img rows), not once per ray. */
typedef struct {
    const float* embed;              /* [num_img][64] */
    const float* w_trunk[4];         /* [64][64] */
    const float* b_trunk[4];
    const float *w_rb, *b_rb, *w_vb, *b_vb, *w_wb, *b_wb;   /* branches [32][64] */
    const float *w_r, *b_r, *w_v, *b_v;                     /* heads [3M][32] */
    const float *w_w, *b_w;                                 /* [M+1][32] */
} lush_rbk_params;
typedef struct {
    float* embed;
    float* w_trunk[4];
    float* b_trunk[4];
    float *w_rb, *b_rb, *w_vb, *b_vb, *w_wb, *b_wb;
    float *w_r, *b_r, *w_v, *b_v;
    float *w_w, *b_w;
} lush_rbk_grads;
#define LUSH_RBK_ACT_STRIDE 512      /* floats per image in `acts` */
#define LUSH_RBK_RVW_STRIDE 32       /* r(12) v(12) w(5) pad(3) */
#define LUSH_RBK_RVW_OFFSET 480      /* acts[i][480..511]: lush_rbk_mlp_fwd leaves them ZERO, so the gradients w.r.t. r, v, w of image
                                      * i may be accumulated right there (d_rvw = acts + 480, rvw_stride = 512): no buffer of their
                                      * own, no zero-fill launch */
/* acts [num_img][512] (hidden activations + r, v, w; every element written, [464..511] as zeros). */
int lush_rbk_mlp_fwd(const lush_rbk_params* p, int num_img, int M, float window, float* acts,
                     lush_stream_t stream);
/* d_rvw [num_img][rvw_stride >= 32] = gradients w.r.t. r(12), v(12), normalised w(5) in the first 29 floats of a row; they are
 * CONSUMED: the kernel leaves those 29 floats zero (ABI 9), so rows kept in the zero tail of `acts` are zero again for the next
 * lush_rbk_warp_bwd / _ndc_bwd over the same activations (a second backward of a retained graph).
 * `g` is added to when accumulate != 0 (gradient buffers that already hold a slice's contribution), else overwritten (the
 * entry point zeroes it first: the images are split over workgroups and the sums leave by atomics either way).  `scratch` is
 * unused since ABI 7 and may be NULL. */
int lush_rbk_mlp_bwd(const lush_rbk_params* p, int num_img, int M, float window, const float* acts,
                     float* d_rvw, int rvw_stride, const lush_rbk_grads* g, float* scratch, int accumulate,
                     lush_stream_t stream);
/* rbk_warp (models/lushnerf.py:75-98) + SE3Field.warp (utils/rigid_warping.py:20-140):
 * rays [N][3][2], idx [N] int64 -> new_rays [N*(M+1)][3][2] (slot 0 = input ray),
 * ccw [N][M+1]. */
int lush_rbk_warp_fwd(const float* rays, const int64_t* idx, int N, int M, const float* acts,
                      float* new_rays, float* ccw, lush_stream_t stream);
/* mask [N] (uint8) or NULL: rays whose mask is 0 pass no gradient through
 * new_rays (the allkernel torch.where(..., x, x.detach()), models/lushnerf.py:641-643).
 * d_rvw [num_img][rvw_stride] accumulate (zero before the first call of a step); drays [N][3][2]
 * overwritten, may be NULL. */
int lush_rbk_warp_bwd(const float* rays, const int64_t* idx, int N, int M, const float* acts,
                      const float* dnew_rays, const float* dccw, const uint8_t* mask,
                      float* d_rvw, int rvw_stride, float* drays, lush_stream_t stream);
/* The two above and lush_pack_rays_fwd / _bwd in ONE kernel per direction (SURVEY.md section 7.2 `rbk_warp_ndc`):
 * Rigid_Blurring_Kernel.forward's warp (models/lushnerf.py:75-98, utils/rigid_warping.py:20-140) followed by the head of
 * render_train_scene (:772-795; ndc_rays, utils/run_lushnerf_helpers.py:542-562) for the M + 1 rays of every input ray --
 * the warped rays never exist in memory -- and the same head for the input ray alone, which is what render_train_noise
 * (:827-850) marches: rays [N][3][2], idx [N] -> batch [N*(M+1)][11] (row n*(M+1) = the input ray), ccw [N][M+1],
 * batch0 [N][11] (may be NULL).  cx, cy, near, far as lush_pack_rays_fwd. */
int lush_rbk_warp_ndc_fwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, int ndc, float cx, float cy,
                          float near, float far, float* batch, float* ccw, float* batch0, lush_stream_t stream);
/* dbatch [N*(M+1)][11] (NULL: no gradient through the rays) and dccw [N][M+1] (NULL: none through the weights) -> d_rvw
 * (accumulate, as lush_rbk_warp_bwd), drays [N][3][2] (overwritten; may be NULL). */
int lush_rbk_warp_ndc_bwd(const float* rays, const int64_t* idx, int N, int M, const float* acts, int ndc, float cx, float cy,
                          const float* dbatch, const float* dccw, const uint8_t* mask, float* d_rvw, int rvw_stride,
                          float* drays, int num_img, lush_stream_t stream);       /* num_img: rows of acts / d_rvw; idx < num_img */

/* ------------------------------------------------------- blur mix and tone map
 * rbk_weighted_sum (models/lushnerf.py:100-116): x [N*M][C], ccw [N][M] -> y [N][C]. */
int lush_wsum_fwd(const float* x, const float* ccw, int N, int M, int C, float* y, lush_stream_t stream);
int lush_wsum_bwd(const float* x, const float* ccw, int N, int M, int C, const float* dy,
                  float* dx, float* dccw, lush_stream_t stream);          /* every element of dx, dccw written (no accumulation: ABI 6) */
/* y = (add ? x + 0.1*sigmoid(nraw) : x) ** (1/2.2) when gamma, else without the
 * power (ToneMapping 'gamma'/'none', utils/run_lushnerf_helpers.py:164-174, and
 * models/lushnerf.py:649, 654).  x [n][3]; nraw [n][3] or NULL. */
int lush_tonemap_fwd(const float* x, const float* nraw, int n, int gamma, float* y, lush_stream_t stream);
int lush_tonemap_bwd(const float* x, const float* nraw, int n, int gamma, const float* dy,
                     float* dx, float* dnraw, lush_stream_t stream);      /* every element written */
/* y = 0.1*sigmoid(x), models/lushnerf.py:649, 660. */
int lush_noise_act_fwd(const float* x, int n, float* y, lush_stream_t stream);
int lush_noise_act_bwd(const float* x, int n, const float* dy, float* dx, lush_stream_t stream); /* every element written */
/* The tail of NeRFAll.forward's training branch in ONE kernel per direction (SURVEY.md section 7.2 `blur_mix_tonemap`;
 * models/lushnerf.py:644-654, 100-116; utils/run_lushnerf_helpers.py:164-174): with s = sum_m ccw[n][m] rgb[n*M1+m],
 * s0 the same on rgb0 and nz = 0.1 sigmoid(nraw):  blur = tm(s + nz), blur0 = tm(s0 + nz), noise = nz, sharp = tm(s),
 * sharp0 = tm(s0); tm = x ** (1/2.2) when gamma, else identity.  rgb, rgb0 [N*M1][3]; ccw [N][M1]; nraw [N][>=3] with row
 * stride nraw_ld floats (3, or 4 for the noise MLP's raw output read in place); outputs [N][3]. */
int lush_blur_mix_fwd(const float* rgb, const float* rgb0, const float* ccw, const float* nraw, int nraw_ld, int N, int M1,
                      int gamma, float* blur, float* blur0, float* noise, float* sharp, float* sharp0, lush_stream_t stream);
/* g_* [N][3]: gradients of the five outputs, any of them NULL (= 0).  d_rgb, d_rgb0 [N*M1][3], d_ccw [N][M1], d_nraw [N][4]
 * (column 3 = 0: the row is the noise MLP's d_raw as lush_mlp_bwd takes it): every element written. */
int lush_blur_mix_bwd(const float* rgb, const float* rgb0, const float* ccw, const float* nraw, int nraw_ld, int N, int M1, int gamma,
                      const float* g_blur, const float* g_blur0, const float* g_noise, const float* g_sharp,
                      const float* g_sharp0, float* d_rgb, float* d_rgb0, float* d_ccw, float* d_nraw, lush_stream_t stream);
/* Training loss of run_lushnerf.py:652-661: sum over the two colours of
 * 0.5*MSE + 0.5*L1 against target [n][3], times `scale` (the share of a micro-batch in the step's
 * mean; 1 for the plain loss).  ga / gb = d loss / d a, d loss / d b (overwritten).
 * work == NULL: loss[0] accumulates (zero it first).  work != NULL: 2 floats of caller-owned scratch, ZERO before the first
 * call and left zero by every call; loss[0] is then WRITTEN (no zero-fill launch per call).
 * gb == NULL (allowed only with a == b: no fine pass, the reference's rgb0 = rgb): ga receives the sum of the two gradients. */
int lush_loss_fwd_bwd(const float* a, const float* b, const float* target, int n, float scale, float* loss,
                      float* ga, float* gb, float* work, lush_stream_t stream);

/* ----------------------------------------------------------------- random draws
 * The four draws of one march in the reference's shapes -- torch.rand [R][Ns] (models/lushnerf.py:515),
 * torch.randn_like [R][Ns-1] (:322), torch.rand [R][Ni] (utils/run_lushnerf_helpers.py:578), torch.randn_like
 * [R][Ns+Ni-1] (:322) -- from one Philox4x32-10 launch; any pointer may be NULL (that draw is not made).
 * (seed, offset) select the stream: the caller advances offset per call.  Uniforms lie in [0, 1). */
int lush_draws(unsigned long long seed, unsigned long long offset, float* t_rand, long long n_t, float* noise_c,
               long long n_c, float* u, long long n_u, float* noise_f, long long n_f, lush_stream_t stream);

/* ------------------------------------------------------------------- the MLPs
 * Embedder + NeRF.forward / NeRF_Noise.forward behind NeRFAll.mlpforward /
 * mlpforward_noise (utils/run_lushnerf_helpers.py:334-344, 394-423, 483-512;
 * models/lushnerf.py:234-293).
 * net 0 = NeRF (D=8, W=256, skip after layer 4), net 1 = NeRF_Noise (D=4, W=128).
 * planes = bf16 planes per operand: 1 plain bf16, 2 parity mode (~2^-17), 3 ~fp32. */
typedef struct {
    const float* w[8];
    const float* b[8];
    const float *w_feat, *b_feat, *w_alpha, *b_alpha, *w_views, *b_views, *w_rgb, *b_rgb;
} lush_mlp_params;
typedef struct {
    float* w[8];
    float* b[8];
    float *w_feat, *b_feat, *w_alpha, *b_alpha, *w_views, *b_views, *w_rgb, *b_rgb;
} lush_mlp_grads;

/* Kernel variants.  The MLP entry points take a bit mask `variant`; 0 is the product's choice (what bench.py
 * measures).  The bits select an older kernel for the same work, for A/B timing and for the tests that check
 * that the variants agree; nothing in the library reads the environment. */
#define LUSH_VARIANT_FWD_HALF 1      /* one fp16 plane forward: two 128-point workgroups per CU (round 2) instead of 64 points per wave */
#define LUSH_VARIANT_FWD_512 2       /* ... the one-workgroup 32-points-per-wave kernel (round 1) */
#define LUSH_VARIANT_BWD_512 4       /* fp16 gradient chain: the one-workgroup 32-points-per-wave kernel (round 1) */
#define LUSH_VARIANT_HEAD_KERNEL 8   /* one-plane backward: K<=3 head gradients by their own kernel instead of riding in the grouped launch */
#define LUSH_VARIANT_BWD_HALF 16     /* fp16 gradient chain: two 128-point workgroups per CU (round 2) instead of 64 points per wave */
#define LUSH_VARIANT_PE_ROWS 64      /* one fp16 plane: the forward stashes the encoded rows and the weight gradients read them (round 2),
                                      * instead of 32 bytes per point that the weight-gradient kernel re-encodes; pass the SAME variant word
                                      * to the forward and the backward of a pass */
/* (bit 32 was round 3's LUSH_VARIANT_NO_OVERLAP -- lush_march_bwd no longer uses a second stream -- and is ignored) */
#define LUSH_VARIANT_DW_SPLIT 128    /* weight gradients of a large pass: ONE job per workgroup on a slice sized by the job's cost per point (round 5
                                      * experiment: one drain / flush / refill per workgroup instead of ten, but the launch then lasts as long as
                                      * its slowest job -- 5.5 ms against the walk's 4.1 ms) instead of every workgroup walking
                                      * all the jobs of its slice of the points */
#define LUSH_VARIANT_DENSE_BWD 256   /* lush_march_fwd / _bwd in the headline mode: stash every point in the forward and run the backward over all of them
                                      * (rounds 1-4) instead of re-running the forward and the backward on the live points only (round 5) */
#define LUSH_VARIANT_KERNEL_BITS 0x3DF /* every bit above that selects a kernel; anything else in the word is ignored */

size_t lush_mlp_packed_bytes(int net, int planes);
/* Pack PLANS (ABI 7): the fragments and fp32 blocks of up to 8 (network, plane code) pairs as ONE launch.  A training step
 * re-packs every network once, after the optimiser moved the parameters (the coarse, fine and noise nets in their forward and,
 * where it differs, backward plane code): three launches per network and direction become one per step.
 *   plan = device buffer of lush_pack_plan_bytes(n_jobs) bytes; lush_pack_plan_build fills it from the host (a set-up call: it
 *   SYNCHRONISES, once per model -- parameter and destination addresses are baked in) and returns the grid size in
 *   *launch_blocks; lush_pack_plan_run enqueues the one kernel. */
typedef struct {
    int net, planes, variant;        /* as lush_mlp_pack_for */
    const lush_mlp_params* prm;
    void* packed;                    /* lush_mlp_packed_bytes(net, planes) bytes */
} lush_pack_job;
size_t lush_pack_plan_bytes(int n_jobs);
int lush_pack_plan_build(const lush_pack_job* jobs, int n_jobs, void* plan, size_t plan_bytes, int* launch_blocks);
/* zero_n > 0 (ABI 8): the same launch clears zero_buf[0 .. zero_n) -- the trainer's flat gradient buffer, i.e. the step's zero_grad
 * (run_lushnerf.py:652 optimizer.zero_grad()) without a fill launch of its own. */
int lush_pack_plan_run(const void* plan, int launch_blocks, float* zero_buf, long long zero_n, lush_stream_t stream);
/* Re-pack the fp32 parameters into MFMA fragment order (forward and transposed). */
int lush_mlp_pack(int net, int planes, const lush_mlp_params* prm, void* packed, lush_stream_t stream);
/* ... only the copies the kernels selected by `variant` read (variant < 0: all of them, as lush_mlp_pack). */
int lush_mlp_pack_for(int net, int planes, const lush_mlp_params* prm, void* packed, int variant, lush_stream_t stream);
/* Bytes of the activation stash for P points (forward -> backward) and of the
 * dZ workspace used inside lush_mlp_bwd. */
size_t lush_mlp_stash_bytes(int net, int planes_fwd, int stash_planes, long long P);
size_t lush_mlp_dstash_bytes(int net, int planes, long long P);
/* rays [R][11], z [R][S] -> raw [R*S][4] (rgb raw x3, sigma raw; sigma = 0 for
 * net 1).  stash has lush_mlp_stash_bytes(net, planes, stash_planes, R*S) bytes and receives
 * the first stash_planes (<= planes) bf16 planes of every activation (the backward only needs as
 * many planes as it computes with) plus the encoded inputs.  stash_planes = 0 is inference:
 * the buffer is then only the kernel's gamma-row workspace and nothing is kept for a backward. */
/* LIMIT: R * S < 2^27 points per launch (forward and backward: the kernels address per-point rows by 32-bit byte offsets from
 * a scalar base); more is REFUSED with an error -- split the ray batch (the stash of 2^27 points would be 590 GB anyway). */
int lush_mlp_fwd(int net, int planes, int stash_planes, const float* rays, const float* z, int R, int S,
                 const void* packed, const lush_mlp_params* prm, float* raw, void* stash, int variant,
                 lush_stream_t stream);
/* Backward: draw [R*S][4] -> parameter gradients (accumulate, fp32 atomics) and
 * dpts [R*S][8] = d/dpoint (3), 0, d/dviewdir (3), 0 (overwritten).
 * planes_f = the stash_planes the forward was called with; planes_b <= planes_f is the
 * plane count of the backward arithmetic; packed_b holds planes_b planes. */
int lush_mlp_bwd(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                 const void* packed_b, const lush_mlp_params* prm, const float* draw,
                 const void* stash, void* dstash, const lush_mlp_grads* grads, float* dpts, int variant,
                 lush_stream_t stream);
/* The same in two halves (so each kernel group can be timed / overlapped separately):
 * _chain runs the fused dX chain (writes dstash + dpts); _weights runs the weight-gradient
 * GEMMs over stash x dstash.  lush_mlp_bwd == _chain followed by _weights.
 * With a 1- or 2-plane backward the feature layer (feature_linear, models/lushnerf.py:259-263: no activation) is
 * not stashed at all: _weights accumulates G = dZv^T h_7 and s = sum dZv and derives
 * dW_feature = Wva^T G, db_feature = Wva^T s, dW_views[:, :W] = G Wf^T + s b_f^T, db_views = s from the fp32
 * parameters in `prm` (which it therefore needs). */
int lush_mlp_bwd_chain(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                       const void* packed_b, const lush_mlp_params* prm, const float* draw,
                       const void* stash, void* dstash, float* dpts, int variant, lush_stream_t stream);
int lush_mlp_bwd_weights(int net, int planes_f, int planes_b, int R, int S, const lush_mlp_params* prm,
                         const float* draw, const void* stash, void* dstash, const lush_mlp_grads* grads,
                         int variant, lush_stream_t stream);
/* d rays from d points: pts = o + d*z (models/lushnerf.py:414, 525).  dpts [R*S][8]
 * -> drays [R][11] accumulate (o: 0..2, d: 3..5, viewdir: 8..10). */
int lush_ray_grad_reduce(const float* dpts, const float* z, int R, int S, float* drays,
                         lush_stream_t stream);

/* Live points (round 5).  A sample whose density pre-activation the ReLU of raw2outputs clamps (models/lushnerf.py:313:
 * relu(raw[..., 3] + noise)) has alpha = 0, weight = 0 and d alpha / d raw = 0: its d_raw row is exactly zero and nothing flows
 * back through its MLP evaluation -- with raw_noise_std = 1 and a density near zero, half of all the points.  The one- and two-plane
 * backward of the 8x256 net (plane codes 1, 2, 17 each way) runs on the live points only: lush_live_compact lists them in grid order (live_idx [R*S] int32, the
 * first cnt[0] entries valid), gathers their d_raw rows (draw_c [R*S][4], rows behind the list zeroed) and gives ray r the range
 * [ray_start[r], ray_start[r+1]) of the list (ray_start [R+1]); cnt [2] int32 = {live points, R*S}; aux: lush_live_aux_bytes(R*S).
 * lush_mlp_fwd_live re-runs the forward with the stash on the list (its i-th point = grid point live_idx[i]; no raw output),
 * lush_mlp_bwd_chain_live / lush_mlp_bwd_weights_live are the launches of lush_mlp_bwd_chain / _weights on those cnt[0] points
 * (the count is read on the device: nothing synchronises), lush_ray_grad_reduce_live folds the list's d(point) rows per ray.
 * Zero rows only are skipped: the gradients are those of the launches over all the points (test: tests/test_gpu_parity.py). */
size_t lush_live_aux_bytes(long long P);
int lush_live_compact(const float* draw, int R, int S, int* live_idx, float* draw_c, int* ray_start, int* cnt, void* aux,
                      lush_stream_t stream);
int lush_mlp_fwd_live(int net, int planes, int stash_planes, const float* rays, const float* z, int R, int S, const void* packed,
                      const lush_mlp_params* prm, void* stash, const int* live_idx, const int* live_cnt, int variant,
                      lush_stream_t stream);
int lush_mlp_bwd_chain_live(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                            const void* packed_b, const lush_mlp_params* prm, const float* draw_c, const void* stash, void* dstash,
                            float* dpts, const int* live_idx, const int* live_cnt, int variant, lush_stream_t stream);
int lush_mlp_bwd_weights_live(int net, int planes_f, int planes_b, int R, int S, const lush_mlp_params* prm, const float* draw_c,
                              const void* stash, void* dstash, const lush_mlp_grads* grads, const int* live_cnt, int variant,
                              lush_stream_t stream);
int lush_ray_grad_reduce_live(const float* dpts, const float* z, const int* live_idx, const int* ray_start, int R, float* drays,
                              lush_stream_t stream);

/* ------------------------------------------------------ the march in one call
 * NeRFAll.render_rays_nonoise (models/lushnerf.py:481-583) -- z grid + jitter (:501-523), coarse MLP
 * (mlpforward :234-266), raw2outputs (:296-352), sample_pdf + sort (:544-549, utils/run_lushnerf_helpers.py:566-609),
 * fine MLP, raw2outputs -- and its autograd backward, each as ONE call that enqueues the kernels behind the
 * piecewise entry points above on `stream`, out of one caller-provided workspace (lush_march_workspace_bytes; 256-byte
 * aligned).  The forward leaves what the backward needs (z, raw, weights, packed weights; with LUSH_VARIANT_DENSE_BWD or three
 * planes also the activation stashes -- otherwise the backward fills them for the live points itself, see "Live points") in the
 * workspace: keep it untouched between the two calls.  render_rays (:354-479) is this plus lush_zfixed + lush_mlp_fwd of
 * the noise net; the blur-kernel branch feeds it N (M+1) warped rays (lush_rbk_warp_fwd + lush_pack_rays_fwd). */
typedef struct {
    int R;                         /* marched rays */
    int N_samples, N_importance;   /* N_importance = 0: coarse pass only */
    float perturb, raw_noise_std;  /* > 0: the matching draws are used */
    int white_bkgd, lindisp;
    float near_mask;               /* eval only: render_rmnearplane / 128 (models/lushnerf.py:331-335); < 0 = off */
    int planes_fwd;                /* plane code of the forward: 1..3 bf16 planes, 17 = one fp16 plane */
    int planes_bwd;                /* plane code of the backward; 0 = inference (nothing is kept, lush_march_bwd refuses) */
    int variant;                   /* LUSH_VARIANT_* bits, 0 = the product's choice */
    int same_net;                  /* the fine pass evaluates the coarse parameters (mlp_fine is None, models/lushnerf.py:214) */
    /* ABI 7.  Weights already packed by the caller (lush_mlp_pack_for / a pack plan) for plane code planes_fwd, or NULL: the
     * forward then packs into the workspace itself, per call.  A training step packs every network ONCE (lush_pack_plan_run)
     * and hands the buffers to each of its marches (micro-batches, the consistency branch).  With planes_bwd != planes_fwd the
     * backward needs fragments of its own plane code: packed_bwd_* (NULL: packed inside lush_march_bwd). */
    const void *packed_coarse, *packed_fine, *packed_bwd_coarse, *packed_bwd_fine;
} lush_march_cfg;
/* The random draws of a march in the reference's shapes (see lush_draws); NULL = that draw is off. */
typedef struct { const float *t_rand, *noise_c, *u, *noise_f; } lush_march_draws;
/* Caller-owned outputs: rgb [R][3], depth [R], acc [R], density [R][S(+Ni)-1] of the final pass; with N_importance > 0
 * also the coarse pass's rgb0, depth0, acc0, density0 [R][S-1] and z_std [R] (:465). */
typedef struct { float *rgb, *depth, *acc, *density, *rgb0, *depth0, *acc0, *density0, *z_std; } lush_march_out;
/* d loss / d (rgb_map, depth_map, acc_map, rgb0, depth0, acc0); NULL = zero.  A pass none of whose outputs
 * received a gradient is skipped (the coarse net of the consistency branch, models/lushnerf.py:949-989). */
typedef struct { const float *rgb, *depth, *acc, *rgb0, *depth0, *acc0; } lush_march_gout;

/* 0 = bad configuration, which includes R * (N_samples + N_importance) >= 2^27: no launch behind the march takes that many points
 * (see lush_mlp_fwd), so the march is refused when it is sized, not half way through its backward.  A live-point march with a fine
 * pass keeps ONE stash region for both passes (the coarse pass's is the head of the fine pass's: LUSH_VIEW_STASH_COARSE / _FINE then
 * name overlapping bytes; the dense form, which is what a caller that wants to read the stash runs, keeps two). */
size_t lush_march_workspace_bytes(const lush_march_cfg* cfg);
/* Where the forward left its by-products inside the workspace (byte offset, size): */
#define LUSH_VIEW_Z 0             /* z_vals of the final pass [R][S(+Ni)], sorted */
#define LUSH_VIEW_RAW 1           /* raw of the final pass [R][S(+Ni)][4] */
#define LUSH_VIEW_WEIGHTS 2       /* compositing weights of the final pass [R][S(+Ni)] */
#define LUSH_VIEW_Z_COARSE 3      /* z_vals of the coarse pass [R][S] */
#define LUSH_VIEW_STASH_COARSE 4  /* activation stash of the coarse / fine MLP evaluation (tests: lush_debug_stash_layout); after a live-point
                                   * backward: the stash of the pass's LIVE points in list order */
#define LUSH_VIEW_STASH_FINE 5
#define LUSH_VIEW_LIVE_COUNTS 6   /* int32 [4] after lush_march_bwd of a live-point configuration: {live, all} points of the fine pass (or the only pass), then of the coarse pass */
int lush_march_view(const lush_march_cfg* cfg, int which, size_t* offset, size_t* bytes);
/* rays [R][11]; flags: the numerical-fault word (may be NULL).  fine may be NULL when same_net or N_importance == 0. */
int lush_march_fwd(const lush_march_cfg* cfg, const float* rays, const lush_mlp_params* coarse, const lush_mlp_params* fine,
                   const lush_march_draws* draws, const lush_march_out* out, void* workspace, int* flags, lush_stream_t stream);
/* Parameter gradients are ADDED to g_coarse / g_fine (fp32 atomics; same_net: everything goes to g_coarse);
 * drays [R][11] = d loss / d ray batch (columns 0..5, 8..10; 6, 7 zero), WRITTEN by the first pass that runs (ABI 7: the caller
 * no longer zero-fills it; when no output gradient is given at all nothing runs and drays is left untouched); z_samples are
 * detached as in the reference (:546).  The passes run one after the other on `stream`: nothing else is created or used. */
int lush_march_bwd(const lush_march_cfg* cfg, const float* rays, const lush_mlp_params* coarse, const lush_mlp_params* fine,
                   const lush_march_draws* draws, const lush_march_gout* gout, void* workspace, const lush_mlp_grads* g_coarse,
                   const lush_mlp_grads* g_fine, float* drays, lush_stream_t stream);

/* ---------------------------------------------------------------------- Adam
 * torch.optim.Adam step on a flat segment (run_lushnerf.py:368-371, 675-685). */
int lush_adam(float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1,
              float beta2, float eps, int step, float grad_scale, lush_stream_t stream);
/* One launch for up to three CONSECUTIVE segments of one flat buffer: segment s = elements [end(s-1), end(s)), end(-1) = 0, at its
 * own step count steps[s] (torch.optim.Adam keeps one per parameter; parameters without a gradient do not count the step); bit s
 * of `mask` clear = segment s is skipped. */
int lush_adam_multi(float* param, const float* grad, float* m, float* v, long long end0, long long end1, long long end2, int mask,
                    float lr, float beta1, float beta2, float eps, const int* steps, float grad_scale, lush_stream_t stream);

/* ------------------------------------------------------------- step state on the device
 * What a training step otherwise takes from the host as kernel arguments -- the learning rate (run_lushnerf.py:675-685), Adam's
 * step counts (as their bias corrections) and the Philox draw counter -- in lush_step_state_bytes() of device memory, so that a step
 * captured in a HIP graph (every entry point only enqueues on the caller's stream) advances when the graph is replayed:
 * lush_draws_state adds the state's draw counter to `offset` (the caller passes the number of the call inside the step),
 * lush_adam_state reads rate and bias corrections of segment 0..2, lush_step_state_advance ends the step: the draw counter moves by
 * n_draw_calls, the segments in active_mask and global_step by one, the next rate = lrate * 0.1 ** (max(global_step - 1, 0) /
 * decay_steps) (computed in double, as the host does). */
size_t lush_step_state_bytes(void);
int lush_step_state_init(void* state, unsigned long long draw_base, int global_step, const int* adam_steps /* host, 3 */, double lrate,
                         double decay_steps, double beta1, double beta2, lush_stream_t stream);
int lush_step_state_advance(void* state, int n_draw_calls, int active_mask, double lrate, double decay_steps, double beta1,
                            double beta2, lush_stream_t stream);
int lush_draws_state(unsigned long long seed, unsigned long long offset, const void* state, float* t_rand, long long n_t,
                     float* noise_c, long long n_c, float* u, long long n_u, float* noise_f, long long n_f, lush_stream_t stream);
int lush_adam_state(float* param, const float* grad, float* m, float* v, long long n, const void* state, int segment,
                    float beta1, float beta2, float eps, float grad_scale, lush_stream_t stream);
/* The same step for up to three CONSECUTIVE segments of one flat buffer in one launch: segment s covers the elements
 * [end(s-1), end(s)) with end(-1) = 0; bit s of `mask` clear = the segment is skipped (no gradient this step). */
int lush_adam_state_multi(float* param, const float* grad, float* m, float* v, long long end0, long long end1, long long end2,
                          const void* state, int mask, float beta1, float beta2, float eps, float grad_scale, lush_stream_t stream);

/* Test hooks (tests/ only): raw access to a stash array for layer-wise parity. */
int lush_debug_stash_layout(int net, int planes, long long P, long long* offsets /* host, 16 entries */);

#ifdef __cplusplus
}
#endif
#endif
