// lush-march: ONE C-ABI call per ray march: lush_march_fwd / lush_march_bwd are
// NeRFAll.render_rays_nonoise (models/lushnerf.py:481-583) and its autograd backward, composed on the host from the
// kernels behind the piecewise entry points -- z grid + jitter, weight packing, coarse MLP, compositing, sample_pdf +
// merge, fine MLP, compositing -- enqueued on the caller's stream, working out of ONE caller-provided workspace whose
// size is queried first (lush_march_workspace_bytes).  No allocation, no synchronisation, no state between calls: what
// the backward needs (z, raw, weights, the packed weights, the activation stashes) stays in the workspace.
#include "lush_common.h"
#include "lush_host.h"
#include "../../include/lush_march.h"

using namespace lush;

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int nplanes(int c) { return c == PLANES_F16 ? 1 : c; }
// what the forward keeps for the backward (ops.stash_code): an fp16 forward stashes its single fp16 plane, otherwise the
// first min(forward planes, backward planes) bf16 planes
inline int stash_code(int pf, int pb) { return pb == 0 ? 0 : (pf == PLANES_F16 ? PLANES_F16 : (pf < nplanes(pb) ? pf : nplanes(pb))); }

struct Layout {
    size_t zc, wc, rawc, pkc, stashc;            // coarse pass
    size_t zf, zs, wf, rawf, pkf, stashf;        // fine pass (N_importance > 0)
    size_t pkbc, pkbf;                           // packed weights of the backward when its plane code differs
    size_t draw, dstash, dpts;                   // backward scratch, shared by the passes (they run one after the other)
    size_t live_idx, draw_c, ray_start, live_cnt, live_aux;     // live-point backward (round 5): the list, its d_raw rows, per-ray ranges, count, scan scratch
    size_t total;
    size_t stashc_bytes, stashf_bytes;
    bool live;                                   // the backward re-runs the forward on the live points (the forward call keeps no stash)
};

// The march of every one- / two-plane mode (the 64-points-per-wave kernels of (h,h), the 128-point chain kernels of the bf16-plane
// modes) keeps NO stash in its forward: the
// backward lists the points whose d_raw row is non-zero, re-runs the forward with the stash on that list and chains / forms the
// weight gradients on it (include/lush_march.h "Live points").  LUSH_VARIANT_DENSE_BWD keeps the rounds 1-4 form.
inline bool live_mode(const lush_march_cfg* c) {
    return c->planes_bwd != 0 && !(c->variant & LUSH_VARIANT_DENSE_BWD) &&
           mlp_live_kernels(0, c->planes_fwd, c->planes_bwd, c->variant & LUSH_VARIANT_KERNEL_BITS & ~LUSH_VARIANT_DENSE_BWD);
}

bool layout(const lush_march_cfg* c, Layout& L) {
    if (!c || c->R <= 0 || c->N_samples < 2 || c->N_importance < 0) return false;
    const long long R = c->R, S = c->N_samples, Ni = c->N_importance, Sf = S + Ni;
    // (every MLP launch and lush_live_compact take at most 2^27 - 1 points: refused HERE, not by the first launch that meets them --
    // round 5's layout sized a workspace for such a march and its backward then failed half way)
    if (R * Sf >= (1LL << 27)) return false;
    const int pf = c->planes_fwd, pb = c->planes_bwd, sc = stash_code(pf, pb);
    const size_t pk_f = lush_mlp_packed_bytes(0, pf), pk_b = pb ? lush_mlp_packed_bytes(0, pb) : 0;
    if (pk_f == 0 || (pb && pk_b == 0)) return false;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
    L = Layout{};
    L.live = pb != 0 && live_mode(c);
    L.zc = take(R * S * 4); L.wc = take(R * S * 4); L.rawc = take(R * S * 16); L.pkc = take(pk_f);
    L.stashc_bytes = lush_mlp_stash_bytes(0, pf, sc, R * S);
    // Live-point march with a fine pass: ONE stash region for both passes.  Its forward keeps no stash (the region is the
    // inference kernels' scratch), its backward re-runs the forward with the stash pass by pass, fine first, and a pass is done with
    // its stash before the next one starts: the coarse pass's region is the head of the fine pass's (-4.4 KB per coarse point of
    // workspace: 5.8 GB at BASELINE config 2).
    const bool share_stash = L.live && Ni > 0;
    if (!share_stash) L.stashc = take(L.stashc_bytes);
    if (Ni > 0) {
        L.zf = take(R * Sf * 4); L.zs = take(R * Ni * 4); L.wf = take(R * Sf * 4); L.rawf = take(R * Sf * 16);
        L.pkf = c->same_net ? L.pkc : take(pk_f);
        L.stashf_bytes = lush_mlp_stash_bytes(0, pf, sc, R * Sf);
        L.stashf = take(L.stashf_bytes);
        if (share_stash) L.stashc = L.stashf;
    }
    if (pb) {
        const long long Pmax = R * (Ni > 0 ? Sf : S);
        L.pkbc = pb == pf ? L.pkc : take(pk_b);
        L.pkbf = Ni > 0 ? (pb == pf ? L.pkf : (c->same_net ? L.pkbc : take(pk_b))) : 0;
        L.draw = take(Pmax * 16);
        L.dstash = take(lush_mlp_dstash_bytes(0, pb, Pmax));
        L.dpts = take(Pmax * 32);
        if (L.live) {
            L.live_idx = take(Pmax * 4);
            L.draw_c = take(Pmax * 16);
            L.ray_start = take(((size_t)R + 1) * 4);
            L.live_cnt = take(256);
            L.live_aux = take(lush_live_aux_bytes(Pmax));
        }
    }
    L.total = off;
    return true;
}

}  // namespace

extern "C" {

size_t lush_march_workspace_bytes(const lush_march_cfg* cfg) {
    Layout L;
    return layout(cfg, L) ? L.total : 0;
}

int lush_march_view(const lush_march_cfg* cfg, int which, size_t* offset, size_t* bytes) {
    Layout L;
    if (!layout(cfg, L) || !offset || !bytes) return set_error("lush_march_view: bad configuration");
    const long long R = cfg->R, S = cfg->N_samples, Ni = cfg->N_importance, Sf = S + Ni;
    const bool fine = Ni > 0;
    switch (which) {
        case LUSH_VIEW_Z: *offset = fine ? L.zf : L.zc; *bytes = R * (fine ? Sf : S) * 4; return 0;
        case LUSH_VIEW_RAW: *offset = fine ? L.rawf : L.rawc; *bytes = R * (fine ? Sf : S) * 16; return 0;
        case LUSH_VIEW_WEIGHTS: *offset = fine ? L.wf : L.wc; *bytes = R * (fine ? Sf : S) * 4; return 0;
        case LUSH_VIEW_Z_COARSE: *offset = L.zc; *bytes = R * S * 4; return 0;
        case LUSH_VIEW_STASH_COARSE: *offset = L.stashc; *bytes = L.stashc_bytes; return 0;
        case LUSH_VIEW_STASH_FINE:
            if (!fine) return set_error("lush_march_view: no fine pass");
            *offset = L.stashf; *bytes = L.stashf_bytes; return 0;
        case LUSH_VIEW_LIVE_COUNTS:
            if (!L.live) return set_error("lush_march_view: this configuration's backward runs over all the points");
            *offset = L.live_cnt; *bytes = 16; return 0;
        default: return set_error("lush_march_view: unknown view");
    }
}

int lush_march_fwd(const lush_march_cfg* cfg, const float* rays, const lush_mlp_params* coarse, const lush_mlp_params* fine,
                   const lush_march_draws* draws, const lush_march_out* out, void* workspace, int* flags, lush_stream_t st) {
    Layout L;
    if (!layout(cfg, L)) return set_error("lush_march_fwd: bad configuration");
    if (!rays || !coarse || !out || !workspace) return set_error("lush_march_fwd: rays, coarse parameters, outputs and workspace are required");
    const int R = cfg->R, S = cfg->N_samples, Ni = cfg->N_importance, Sf = S + Ni;
    const bool two = Ni > 0;
    if (two && !cfg->same_net && !fine) return set_error("lush_march_fwd: the fine parameters are required");
    if (!out->rgb || !out->depth || !out->acc || !out->density) return set_error("lush_march_fwd: rgb, depth, acc, density outputs are required");
    if (two && (!out->rgb0 || !out->depth0 || !out->acc0 || !out->density0 || !out->z_std)) return set_error("lush_march_fwd: the coarse outputs and z_std are required when N_importance > 0");
    const lush_mlp_params* pfine = cfg->same_net ? coarse : fine;
    char* w = (char*)workspace;
    const int pf = cfg->planes_fwd, var = cfg->variant;
    const int sc = L.live ? 0 : stash_code(pf, cfg->planes_bwd);      // live-point backward: the forward keeps raw, z, weights only
    const float* t_rand = draws && cfg->perturb > 0.f ? draws->t_rand : nullptr;
    const float* noise_c = draws && cfg->raw_noise_std > 0.f ? draws->noise_c : nullptr;
    const float* u = draws && cfg->perturb > 0.f ? draws->u : nullptr;
    const float* noise_f = draws && cfg->raw_noise_std > 0.f ? draws->noise_f : nullptr;
    float* zc = (float*)(w + L.zc);
    int rc = lush_zgrid(rays, R, S, cfg->lindisp, t_rand, zc, st);
    if (rc) return rc;
    // weights: packed by the caller once per step (cfg->packed_*), or here into the workspace
    const void* pkc = cfg->packed_coarse;
    if (!pkc) {
        rc = lush_mlp_pack_for(0, pf, coarse, w + L.pkc, cfg->variant, st);
        if (rc) return rc;
        pkc = w + L.pkc;
    }
    rc = lush_mlp_fwd(0, pf, sc, rays, zc, R, S, pkc, coarse, (float*)(w + L.rawc), w + L.stashc, var, st);
    if (rc) return rc;
    // the coarse pass's results are rgb0 .. when a fine pass follows, else the final ones
    rc = lush_composite_fwd((const float*)(w + L.rawc), zc, rays, R, S, noise_c, cfg->raw_noise_std, cfg->near_mask, cfg->white_bkgd,
                            two ? out->rgb0 : out->rgb, two ? out->depth0 : out->depth, two ? out->acc0 : out->acc, (float*)(w + L.wc),
                            two ? out->density0 : out->density, flags, two ? LUSH_FAULT_COARSE_SHIFT : 0, st);
    if (rc || !two) return rc;
    float* zf = (float*)(w + L.zf);
    rc = lush_sample_merge(zc, (const float*)(w + L.wc), R, S, Ni, u, zf, (float*)(w + L.zs), out->z_std, flags, st);
    if (rc) return rc;
    const void* pkf = cfg->same_net ? pkc : cfg->packed_fine;
    if (!pkf) {
        rc = lush_mlp_pack_for(0, pf, pfine, w + L.pkf, cfg->variant, st);
        if (rc) return rc;
        pkf = w + L.pkf;
    }
    rc = lush_mlp_fwd(0, pf, sc, rays, zf, R, Sf, pkf, pfine, (float*)(w + L.rawf), w + L.stashf, var, st);
    if (rc) return rc;
    return lush_composite_fwd((const float*)(w + L.rawf), zf, rays, R, Sf, noise_f, cfg->raw_noise_std, cfg->near_mask, cfg->white_bkgd,
                              out->rgb, out->depth, out->acc, (float*)(w + L.wf), out->density, flags, 0, st);
}

int lush_march_bwd(const lush_march_cfg* cfg, const float* rays, const lush_mlp_params* coarse, const lush_mlp_params* fine,
                   const lush_march_draws* draws, const lush_march_gout* g, void* workspace, const lush_mlp_grads* g_coarse,
                   const lush_mlp_grads* g_fine, float* drays, lush_stream_t st) {
    Layout L;
    if (!layout(cfg, L)) return set_error("lush_march_bwd: bad configuration");
    if (cfg->planes_bwd == 0) return set_error("lush_march_bwd: the forward ran as inference (planes_bwd = 0): nothing was kept");
    if (!rays || !coarse || !g || !workspace || !drays) return set_error("lush_march_bwd: rays, parameters, output gradients, workspace and drays are required");
    const int R = cfg->R, S = cfg->N_samples, Ni = cfg->N_importance, Sf = S + Ni;
    const bool two = Ni > 0;
    const lush_mlp_params* pfine = cfg->same_net ? coarse : fine;
    const lush_mlp_grads* gfine = cfg->same_net ? g_coarse : g_fine;
    char* w = (char*)workspace;
    const int pf = cfg->planes_fwd, pb = cfg->planes_bwd, sc = stash_code(pf, pb), var = cfg->variant & LUSH_VARIANT_KERNEL_BITS;
    const float* noise_c = draws && cfg->raw_noise_std > 0.f ? draws->noise_c : nullptr;
    const float* noise_f = draws && cfg->raw_noise_std > 0.f ? draws->noise_f : nullptr;
    // One pass = compositing backward, (re-pack), gradient chain, d(point) -> d(ray), weight gradients; the fine pass first,
    // then the coarse one (which does not depend on it: z_samples are detached, models/lushnerf.py:546), all on the caller's
    // stream out of one backward scratch.  (Round 3 ran the fine pass's weight gradients on a second stream beside the coarse
    // chain, each on part of the chip: both kernels slowed down side by side -- HBM and power are shared -- and the step
    // gained 0.4 ms on one box and lost 0.3 on two others, the driver's among them; removed in round 4)
    bool first_pass = true;
    struct Scratch { float* draw; char* dstash; float* dpts; };
    const Scratch x{(float*)(w + L.draw), w + L.dstash, (float*)(w + L.dpts)};
    // fragments of the backward's plane code: the caller's (cfg->packed_*), else the forward's copy in the workspace when the
    // codes agree, else packed here
    auto chain = [&](const lush_mlp_params* prm, size_t zoff, size_t rawoff, size_t stashoff, const void* pk_ready, size_t pkboff, bool repack,
                     int Sp, const float* noise, const float* g_rgb, const float* g_depth, const float* g_acc, const void* pk_fwd, int slot) -> int {
        const float* z = (const float*)(w + zoff);
        // the compositing backward also leaves the per-workgroup maxima of |d_raw| (in the d(point) array, which the chain only
        // writes afterwards) for the fp16 chain's loss scale, zeroes the weight-gradient scratch and, in the march's first pass,
        // writes d(ray) whole: no pass over d_raw for the scale, no memsets, no zero-fill of drays by the caller
        float *scale4 = nullptr, *zero_buf = nullptr;
        long long zero_n = 0;
        if (!mlp_dstash_header(0, pb, (long long)R * Sp, x.dstash, &scale4, &zero_buf, &zero_n)) return set_error("lush_march_bwd: bad backward plane code");
        float* block_max = scale4 ? x.dpts : nullptr;
        int rc = lush_composite_bwd((const float*)(w + rawoff), z, rays, R, Sp, noise, cfg->raw_noise_std, cfg->near_mask, cfg->white_bkgd,
                                    g_rgb, g_depth, g_acc, x.draw, drays, block_max, zero_buf, zero_n, first_pass ? 1 : 0, st);
        first_pass = false;
        if (rc) return rc;
        if (scale4) {
            rc = lush_loss_scale(block_max, lush_composite_bwd_blocks(R), scale4, st);
            if (rc) return rc;
        }
        const void* pk = pk_ready;
        if (!pk) {
            if (repack) {       // the backward computes with another plane code than the forward: its own fragments
                rc = lush_mlp_pack_for(0, pb, prm, w + pkboff, cfg->variant, st);
                if (rc) return rc;
            }
            pk = w + pkboff;
        }
        if (L.live) {
            // the live points of this pass: list, gathered d_raw rows, per-ray ranges (three small launches); the forward once
            // more, WITH the stash, on the list; then chain and d(ray) on `cnt` points (read on the device)
            int* lidx = (int*)(w + L.live_idx);
            int* lcnt = (int*)(w + L.live_cnt) + 2 * slot;      // {live points, points} of the fine (slot 0) / coarse (slot 1) pass: LUSH_VIEW_LIVE_COUNTS
            float* draw_c = (float*)(w + L.draw_c);
            rc = lush_live_compact(x.draw, R, Sp, lidx, draw_c, (int*)(w + L.ray_start), lcnt, w + L.live_aux, st);
            if (rc) return rc;
            const void* pkf = pk_fwd;            // forward fragments: the caller's buffer or the forward call's copy in the workspace
            rc = lush_mlp_fwd_live(0, pf, sc, rays, z, R, Sp, pkf, prm, w + stashoff, lidx, lcnt, var, st);
            if (rc) return rc;
            rc = mlp_bwd_chain_prepared(0, sc, pb, rays, z, R, Sp, pk, prm, draw_c, w + stashoff, x.dstash, x.dpts, var, st, lidx, lcnt);
            if (rc) return rc;
            return lush_ray_grad_reduce_live(x.dpts, z, lidx, (const int*)(w + L.ray_start), R, drays, st);
        }
        rc = mlp_bwd_chain_prepared(0, sc, pb, rays, z, R, Sp, pk, prm, x.draw, w + stashoff, x.dstash, x.dpts, var, st);
        if (rc) return rc;
        return lush_ray_grad_reduce(x.dpts, z, R, Sp, drays, st);
    };
    auto weights = [&](const lush_mlp_params* prm, const lush_mlp_grads* gr, size_t stashoff, int Sp, int slot) -> int {
        if (L.live) return mlp_bwd_weights_prepared(0, sc, pb, R, Sp, prm, (const float*)(w + L.draw_c), w + stashoff, x.dstash, gr, var, st, (const int*)(w + L.live_cnt) + 2 * slot);
        return mlp_bwd_weights_prepared(0, sc, pb, R, Sp, prm, x.draw, w + stashoff, x.dstash, gr, var, st);
    };
    const bool any_main = g->rgb || g->depth || g->acc;
    const bool any_c = two ? (g->rgb0 || g->depth0 || g->acc0) : any_main;
    if ((two && any_main && !gfine) || (any_c && !g_coarse))
        return set_error("lush_march_bwd: gradient buffers of a pass that received output gradients are required");
    int rc = 0;
    bool packed_b_c = false;
    // (pb == pf: the forward's fragments serve -- the caller's buffer if it packed, else the workspace copy at L.pkbc / L.pkbf = L.pkc / L.pkf)
    const void* ready_c = pb == pf ? cfg->packed_coarse : cfg->packed_bwd_coarse;
    const void* ready_f = cfg->same_net ? ready_c : (pb == pf ? cfg->packed_fine : cfg->packed_bwd_fine);
    if (two && any_main) {
        rc = chain(pfine, L.zf, L.rawf, L.stashf, ready_f, L.pkbf, pb != pf, Sf, noise_f, g->rgb, g->depth, g->acc,
                   cfg->same_net ? (cfg->packed_coarse ? cfg->packed_coarse : (const void*)(w + L.pkc)) : (cfg->packed_fine ? cfg->packed_fine : (const void*)(w + L.pkf)), 0);
        if (rc) return rc;
        packed_b_c = cfg->same_net && pb != pf;     // the shared net's backward fragments are packed now
        rc = weights(pfine, gfine, L.stashf, Sf, 0);
        if (rc) return rc;
    }
    if (any_c) {
        rc = chain(coarse, L.zc, L.rawc, L.stashc, ready_c, L.pkbc, pb != pf && !packed_b_c, S, noise_c, two ? g->rgb0 : g->rgb,
                   two ? g->depth0 : g->depth, two ? g->acc0 : g->acc, cfg->packed_coarse ? cfg->packed_coarse : (const void*)(w + L.pkc), two ? 1 : 0);
        if (rc) return rc;
        rc = weights(coarse, g_coarse, L.stashc, S, two ? 1 : 0);
    }
    return rc;
}

}  // extern "C"
