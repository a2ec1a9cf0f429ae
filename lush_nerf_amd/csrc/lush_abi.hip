// lush-march: C ABI of the fused MLP path (pack plan, stash layout, launch order).
#include "lush_common.h"
#include "lush_host.h"
#include <cstring>
#include <vector>
#include "../../include/lush_march.h"

using namespace lush;

namespace {

struct NetInfo { int HW, NL, SKIP, HV, NRB, total_entries, f32_total; };
template <class N> NetInfo info_of() { return {N::HW, N::NL, N::SKIP, N::HV, N::NRB, N::total_entries, N::f32_total}; }
bool net_info(int net, NetInfo& o) {
    if (net == 0) { o = info_of<NetNerf>(); return true; }
    if (net == 1) { o = info_of<NetNoise>(); return true; }
    return false;
}

// plane code -> number of 16-bit planes stored / computed with
inline bool code_ok(int c) { return (c >= 1 && c <= 3) || c == PLANES_F16; }
inline int nplanes(int c) { return c == PLANES_F16 ? 1 : c; }
constexpr int PT_PAD = 256;     // point arrays are padded to the largest tile (mlp_wide_fwd_kernel: 256 points)
inline long long pad_pts(long long P) { return (P + PT_PAD - 1) / PT_PAD * PT_PAD; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Byte offsets inside the forward stash: [ReLU masks][h_0..][feature][views hidden] with
// `sp` (stash) planes, then the gamma rows with `pf` (forward) planes -- the forward itself
// re-reads them.  sp == 0 (inference) leaves only the gamma rows.
struct StashLayout {
    size_t mask, mask_dummy, pe, xd, h[NET_MAX_LAYERS], feat, hv, total;
    long long Ppad;
};
StashLayout stash_layout(const NetInfo& n, int pf, int sp, long long P) {
    StashLayout L{};
    L.Ppad = pad_pts(P);
    size_t off = 0;
    L.mask = off; off += sp ? al256((size_t)(L.Ppad / 32) * (n.NL + 1) * n.NRB * 16 * 8) : 0;
    L.mask_dummy = off; off += sp ? 4096 : 0;
    for (int l = 0; l < n.NL; ++l) { L.h[l] = off; off += al256((size_t)sp * L.Ppad * n.HW * 2); }
    // the grouped weight-gradient launch (1 and 2 planes) takes the feature layer's gradients from dZv^T h_{NL-1}
    // (FeatFactorArgs): the feature activations are kept for the three-plane reference mode only
    L.feat = off; off += sp >= 3 ? al256((size_t)sp * L.Ppad * n.HW * 2) : 0;
    L.hv = off;   off += al256((size_t)sp * L.Ppad * n.HV * 2);
    L.xd = off;   off += sp ? al256((size_t)L.Ppad * 32) : 0;      // points and view directions (in front of the gamma rows: independent of pf)
    L.pe = off;   off += al256((size_t)pf * L.Ppad * PE_ROW * 2);
    L.total = off;
    return L;
}
struct DStashLayout { size_t scale, fac, dz[NET_MAX_LAYERS], dfeat, dzv, total; };
DStashLayout dstash_layout(const NetInfo& n, int ns, long long P) {
    DStashLayout L{};
    const long long Ppad = pad_pts(P);
    size_t off = 0;
    L.scale = off; off += 256;      // {loss scale, 1/scale, 2 work words} of the fp16 gradient chain
    // 1 and 2 planes: fp32 scratch G = dZv^T h_{NL-1} [HV][HW] then s = sum dZv [HV] (FeatFactorArgs), no d_feature array
    // (one plane: DZV_EXT more rows of G / s for the heads, and the [DZV_EXT][HV] + [DZV_EXT] block of the rgb job)
    L.fac = off; off += ns <= 2 ? al256((size_t)((n.HV + DZV_EXT) * (n.HW + 1) + DZV_EXT * (n.HV + 1)) * 4) : 0;
    for (int l = 0; l < n.NL; ++l) { L.dz[l] = off; off += al256((size_t)ns * Ppad * n.HW * 2); }
    L.dfeat = off; off += ns >= 3 ? al256((size_t)ns * Ppad * n.HW * 2) : 0;
    L.dzv = off;   off += al256((size_t)ns * Ppad * (n.HV + (ns == 1 ? DZV_EXT : 0)) * 2);   // one plane: [dZv | head gradients]
    L.total = off;
    return L;
}

MlpParams to_params(const lush_mlp_params* p) {
    MlpParams q;
    for (int i = 0; i < NET_MAX_LAYERS; ++i) { q.w[i] = p->w[i]; q.b[i] = p->b[i]; }
    q.w_feat = p->w_feat; q.b_feat = p->b_feat; q.w_alpha = p->w_alpha; q.b_alpha = p->b_alpha;
    q.w_views = p->w_views; q.b_views = p->b_views; q.w_rgb = p->w_rgb; q.b_rgb = p->b_rgb;
    return q;
}

// copy 0 = natural rows (tiled 3-plane kernels), copy 1 = chain_row() permutation (chain kernels)
template <class N>
void build_pack_table(const lush_mlp_params* p, PackTable& T, int& blocks, int copy_lo, int copy_hi) {
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, SK = N::SKIP;
    T.n = 0;
    blocks = 0;
    int perm = 0, base = 0;
    auto add = [&](const float* src, int sr, int sk, int rows, int cols, int nrb, int kk, int dst) {
        PackJob& j = T.j[T.n++];
        j.src = src; j.sr = sr; j.sk = sk; j.rows = rows; j.cols = cols; j.nrb = nrb; j.kk = kk; j.perm = perm;
        j.dst_entry = base + dst; j.first_block = blocks;
        blocks += nrb * kk;
    };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    // forward: natural rows for mlp_fwd_kernel, then the row-permuted copy for the chain kernel
    for (int copy = copy_lo; copy <= copy_hi; ++copy) {
        perm = copy; base = copy ? N::fwd2_base : 0;
        add(p->w[0], XV, 1, HW, XV, N::NRB, N::KKX, N::fwd_L(0, false));
        for (int l = 1; l < NL; ++l) {
            if (l == SK) {
                add(p->w[l], XV + HW, 1, HW, XV, N::NRB, N::KKX, N::fwd_L(l, false));
                add(p->w[l] + XV, XV + HW, 1, HW, HW, N::NRB, N::KKH, N::fwd_L(l, true));
            } else {
                add(p->w[l], HW, 1, HW, HW, N::NRB, N::KKH, N::fwd_L(l, true));
            }
        }
        add(p->w_feat, HW, 1, HW, HW, N::NRB, N::KKH, N::fwd_FEAT);
        add(p->w_alpha, HW, 1, 1, HW, 1, N::KKH, N::fwd_ALPHA);
        add(p->w_views, HW + DV, 1, HV, HW, N::NRBV, N::KKH, N::fwd_VA);
        add(p->w_views + HW, HW + DV, 1, HV, DV, N::NRBV, N::KKD, N::fwd_VB);
        add(p->w_rgb, HV, 1, 3, HV, 1, N::KKV, N::fwd_RGB);
    }
    perm = 0; base = 0;
    // transposed: element (row, k) = W[k][c0 + row]; natural rows for mlp_bwd_kernel, permuted copy for the chain kernel
    for (int copy = copy_lo; copy <= copy_hi; ++copy) {
        perm = copy; base = copy ? N::bwd2_base - N::bwd_VAT : 0;
        add(p->w_views, 1, HW + DV, HW, HV, N::NRB, N::KKV, N::bwd_VAT);
        add(p->w_views + HW, 1, HW + DV, DV, HV, 1, N::KVB, N::bwd_VBT);
        add(p->w_feat, 1, HW, HW, HW, N::NRB, N::KKH, N::bwd_FEATT);
        for (int l = NL - 1; l >= 1; --l) {
            if (l == SK) {
                add(p->w[l], 1, XV + HW, XV, HW, 2, N::KKH, N::bwd_LT(l, false));
                add(p->w[l] + XV, 1, XV + HW, HW, HW, N::NRB, N::KKH, N::bwd_LT(l, true));
            } else {
                add(p->w[l], 1, HW, HW, HW, N::NRB, N::KKH, N::bwd_LT(l, true));
            }
        }
        add(p->w[0], 1, XV, XV, HW, 2, N::KKH, N::bwd_LT(0, false));
    }
    perm = 0; base = 0;
}

// Third forward copy (NetT::fwd3_base): half-row stream of mlp_chain_fwd_half_kernel.  Its own table: the kernel
// argument block holds 64 jobs.
template <class N>
void build_pack_table_half(const lush_mlp_params* p, PackTable& T, int& blocks) {
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, SK = N::SKIP, HR = N::HW / 2, NRBH = N::NRB / 2;
    T.n = 0;
    blocks = 0;
    int dst = N::fwd3_base;
    auto add = [&](const float* src, int sr, int rows, int cols, int nrb, int kk) {
        PackJob& j = T.j[T.n++];
        j.src = src; j.sr = sr; j.sk = 1; j.rows = rows; j.cols = cols; j.nrb = nrb; j.kk = kk; j.perm = 1;
        j.dst_entry = dst; j.first_block = blocks;
        blocks += nrb * kk;
        dst += nrb * kk;
    };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    for (int half = 0; half < 2; ++half) add(p->w[0] + (long long)half * HR * XV, XV, HR, XV, NRBH, N::KKX);
    for (int l = 1; l < NL; ++l) {
        const int ld = l == SK ? XV + HW : HW;
        for (int half = 0; half < 2; ++half) {
            const float* w = p->w[l] + (long long)half * HR * ld;
            if (l == SK) {
                add(w, ld, HR, XV, NRBH, N::KKX);
                add(w + XV, ld, HR, HW, NRBH, N::KKH);
            } else {
                add(w, ld, HR, HW, NRBH, N::KKH);
            }
        }
    }
    for (int half = 0; half < 2; ++half) add(p->w_feat + (long long)half * HR * HW, HW, HR, HW, NRBH, N::KKH);
    add(p->w_alpha, HW, 1, HW, 1, N::KKH);
    add(p->w_views, HW + DV, HV, HW, N::NRBV, N::KKH);
    add(p->w_views + HW, HW + DV, HV, DV, N::NRBV, N::KKD);
    add(p->w_rgb, HV, 3, HV, 1, N::KKV);
}

// Fourth forward copy (NetT::fwd4_base): quarter-row stream of mlp_wide_fwd_kernel (lush_mlp_wide.hip).
template <class N>
void build_pack_table_wide(const lush_mlp_params* p, PackTable& T, int& blocks) {
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, SK = N::SKIP, QR = 64;
    T.n = 0;
    blocks = 0;
    int dst = N::fwd4_base;
    auto add = [&](const float* src, int sr, int rows, int cols, int nrb, int kk) {
        PackJob& j = T.j[T.n++];
        j.src = src; j.sr = sr; j.sk = 1; j.rows = rows; j.cols = cols; j.nrb = nrb; j.kk = kk; j.perm = 1;
        j.dst_entry = dst; j.first_block = blocks;
        blocks += nrb * kk;
        dst += nrb * kk;
    };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    for (int q = 0; q < HW / QR; ++q) add(p->w[0] + (long long)q * QR * XV, XV, QR, XV, 2, N::KKX);
    for (int l = 1; l < NL; ++l) {
        const int ld = l == SK ? XV + HW : HW;
        for (int q = 0; q < HW / QR; ++q) {
            const float* w = p->w[l] + (long long)q * QR * ld;
            if (l == SK) {
                add(w, ld, QR, XV, 2, N::KKX);
                add(w + XV, ld, QR, HW, 2, N::KKH);
            } else {
                add(w, ld, QR, HW, 2, N::KKH);
            }
        }
    }
    for (int q = 0; q < HW / QR; ++q) add(p->w_feat + (long long)q * QR * HW, HW, QR, HW, 2, N::KKH);
    add(p->w_alpha, HW, 1, HW, 1, N::KKH);
    for (int q = 0; q < HV / QR; ++q) {
        const float* w = p->w_views + (long long)q * QR * (HW + DV);
        add(w + HW, HW + DV, QR, DV, 2, 4);          // gamma(d) part, K zero-padded to 4 k-blocks (one position)
        add(w, HW + DV, QR, HW, 2, N::KKH);
    }
    add(p->w_rgb, HV, 3, HV, 1, N::KKV);
}

// Fourth transposed copy (NetT::bwd4_base): quarter-row stream of mlp_wide_bwd_kernel (lush_mlp_wide_bwd.hip).
template <class N>
void build_pack_table_wide_bwd(const lush_mlp_params* p, PackTable& T, int& blocks) {
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, SK = N::SKIP, QR = 64;
    T.n = 0;
    blocks = 0;
    int dst = N::bwd4_base;
    // element (row, k) = W[k][c0 + row]: src = W + c0, row stride 1, k stride = W's row length
    auto add = [&](const float* src, int sk, int rows, int cols, int nrb, int kk) {
        PackJob& j = T.j[T.n++];
        j.src = src; j.sr = 1; j.sk = sk; j.rows = rows; j.cols = cols; j.nrb = nrb; j.kk = kk; j.perm = 1;
        j.dst_entry = dst; j.first_block = blocks;
        blocks += nrb * kk;
        dst += nrb * kk;
    };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    for (int q = 0; q < HW / QR; ++q) add(p->w_views + q * QR, HW + DV, QR, HV, 2, N::KKV);
    add(p->w_views + HW, HW + DV, DV, HV, 1, N::KKV);
    for (int q = 0; q < HW / QR; ++q) add(p->w_feat + q * QR, HW, QR, HW, 2, N::KKH);
    for (int l = NL - 1; l >= 1; --l) {
        const int ld = l == SK ? XV + HW : HW;
        for (int q = 0; q < HW / QR; ++q) add(p->w[l] + (l == SK ? XV : 0) + q * QR, ld, QR, HW, 2, N::KKH);
        if (l == SK) add(p->w[l], ld, XV, HW, 2, N::KKH);
    }
    add(p->w[0], XV, XV, HW, 2, N::KKH);
}

// Third transposed copy (NetT::bwd3_base): half-row stream of mlp_chain_bwd_half_kernel.
template <class N>
void build_pack_table_half_bwd(const lush_mlp_params* p, PackTable& T, int& blocks) {
    constexpr int HW = N::HW, HV = N::HV, NL = N::NL, SK = N::SKIP, HR = N::HW / 2, NRBH = N::NRB / 2;
    T.n = 0;
    blocks = 0;
    int dst = N::bwd3_base;
    // element (row, k) = W[k][c0 + row]: src = W + c0, row stride 1, k stride = W's row length
    auto add = [&](const float* src, int sk, int rows, int cols, int nrb, int kk) {
        PackJob& j = T.j[T.n++];
        j.src = src; j.sr = 1; j.sk = sk; j.rows = rows; j.cols = cols; j.nrb = nrb; j.kk = kk; j.perm = 1;
        j.dst_entry = dst; j.first_block = blocks;
        blocks += nrb * kk;
        dst += nrb * kk;
    };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    for (int half = 0; half < 2; ++half) add(p->w_views + half * HR, HW + DV, HR, HV, NRBH, N::KKV);
    add(p->w_views + HW, HW + DV, DV, HV, 1, N::KVB);
    for (int half = 0; half < 2; ++half) add(p->w_feat + half * HR, HW, HR, HW, NRBH, N::KKH);
    for (int l = NL - 1; l >= 1; --l) {
        const int ld = l == SK ? XV + HW : HW;
        if (l == SK) add(p->w[l], ld, XV, HW, 2, N::KKH);
        for (int half = 0; half < 2; ++half) add(p->w[l] + (l == SK ? XV : 0) + half * HR, ld, HR, HW, NRBH, N::KKH);
    }
    add(p->w[0], XV, XV, HW, 2, N::KKH);
}

int dw_splits(long long Ppad) {
    int dev = 0, s = 256;      // one 256x256-tile workgroup per CU of the calling thread's device
    if (current_device_cus(dev, s) != 0) s = 256;
    // every workgroup ends a layer with 256 KB of atomics and starts it with a ring refill: give it at least LUSH_DW_MIN_PTS
    // points (the 4096-point noise net ran 128 workgroups of one tile each: 189 us for 7 tiny GEMMs; 128 points each: 69 us)
#ifndef LUSH_DW_MIN_PTS
#define LUSH_DW_MIN_PTS 128
#endif
    const long long max_s = Ppad / LUSH_DW_MIN_PTS > 0 ? Ppad / LUSH_DW_MIN_PTS : 1;
    if (s > max_s) s = (int)max_s;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" {

size_t lush_mlp_packed_bytes(int net, int planes) {
    NetInfo n;
    if (!net_info(net, n) || !code_ok(planes)) return 0;
    return al256((size_t)n.total_entries * nplanes(planes) * 1024 + (size_t)n.f32_total * 4);
}

// The fragment tables one (net, plane code, variant) needs, in launch order: what lush_mlp_pack / lush_mlp_pack_for launch
// one by one and a pack plan (lush_pack_plan_*) holds as one table.  variant < 0: every copy.
static int collect_pack_tables(int net, int planes, const lush_mlp_params* prm, int variant, std::vector<PackTable>& out,
                               std::vector<int>& out_blocks) {
    if (!code_ok(planes)) return set_error("lush_mlp_pack: planes must be 1..3 or 17 (fp16)");
    if (net != 0 && net != 1) return set_error("lush_mlp_pack: bad net");
    auto push = [&](const PackTable& T, int blocks) { out.push_back(T); out_blocks.push_back(blocks); };
    PackTable T;
    int blocks = 0;
    // the product's kernels for one fp16 plane on the 8x256 net read the two quarter-row streams and the fp32 block only:
    // three launches instead of six (the launch-bound configurations pay for every one of them)
    const int older = LUSH_VARIANT_FWD_HALF | LUSH_VARIANT_FWD_512 | LUSH_VARIANT_BWD_HALF | LUSH_VARIANT_BWD_512;
    if (net == 0 && planes == PLANES_F16 && variant >= 0 && !(variant & older)) {
        build_pack_table_wide<NetNerf>(prm, T, blocks);
        if (blocks != NetNerf::fwd4_len) return set_error("lush_mlp_pack: quarter-row stream length");
        push(T, blocks);
        build_pack_table_wide_bwd<NetNerf>(prm, T, blocks);
        if (blocks != NetNerf::bwd4_len) return set_error("lush_mlp_pack: transposed quarter-row stream length");
        push(T, blocks);
        return 0;
    }
    // only the copies the kernels of this plane count read (the rest of the buffer stays unwritten)
    const bool fc = mlp_fwd_chain_enabled(planes), bc = mlp_bwd_chain_enabled(planes);
    const int lo = (fc && bc) ? 1 : 0, hi = (fc || bc) ? 1 : 0;
    if (net == 0) build_pack_table<NetNerf>(prm, T, blocks, lo, hi);
    else build_pack_table<NetNoise>(prm, T, blocks, lo, hi);
    push(T, blocks);
    if (net == 0 && nplanes(planes) == 1) {      // the half-row streams are read by the one-plane chain kernels only
        build_pack_table_half<NetNerf>(prm, T, blocks);
        push(T, blocks);
        if (planes == PLANES_F16) {              // ... and the half-row backward by the fp16 chain only
            build_pack_table_half_bwd<NetNerf>(prm, T, blocks);
            push(T, blocks);
            build_pack_table_wide<NetNerf>(prm, T, blocks);     // quarter-row forward stream (64 points per wave)
            if (blocks != NetNerf::fwd4_len) return set_error("lush_mlp_pack: quarter-row stream length");
            push(T, blocks);
            build_pack_table_wide_bwd<NetNerf>(prm, T, blocks);     // ... and its transposed twin
            if (blocks != NetNerf::bwd4_len) return set_error("lush_mlp_pack: transposed quarter-row stream length");
            push(T, blocks);
        }
    }
    return 0;
}

int lush_mlp_pack_for(int net, int planes, const lush_mlp_params* prm, void* packed, int variant, lush_stream_t stream) {
    std::vector<PackTable> tables;
    std::vector<int> blocks;
    int rc = collect_pack_tables(net, planes, prm, variant, tables, blocks);
    for (size_t i = 0; i < tables.size() && !rc; ++i) rc = launch_pack(planes, tables[i], blocks[i], packed, (hipStream_t)stream);
    if (rc) return rc;
    return launch_pack_f32(net, nplanes(planes), to_params(prm), packed, (hipStream_t)stream);
}

int lush_mlp_pack(int net, int planes, const lush_mlp_params* prm, void* packed, lush_stream_t stream) {
    return lush_mlp_pack_for(net, planes, prm, packed, -1, stream);
}

// ---- pack plans: every network of a training step in ONE launch ----
size_t lush_pack_plan_bytes(int n_jobs) {
    if (n_jobs < 1 || n_jobs > PLAN_MAX_NETS) return 0;
    return al256(sizeof(PlanHeader) + sizeof(MlpParams) * PLAN_MAX_NETS + sizeof(PlanJob) * (size_t)n_jobs * 260);
}

int lush_pack_plan_build(const lush_pack_job* jobs, int n_jobs, void* plan, size_t plan_bytes, int* launch_blocks) {
    if (!jobs || !plan || !launch_blocks) return set_error("lush_pack_plan_build: jobs, plan and launch_blocks are required");
    if (n_jobs < 1 || n_jobs > PLAN_MAX_NETS) return set_error("lush_pack_plan_build: 1 .. 8 jobs");
    std::vector<PlanJob> pj;
    std::vector<MlpParams> prms(PLAN_MAX_NETS);
    int total = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const lush_pack_job& J = jobs[k];
        if (!J.prm || !J.packed) return set_error("lush_pack_plan_build: a job without parameters or destination");
        std::vector<PackTable> tables;
        std::vector<int> blocks;
        int rc = collect_pack_tables(J.net, J.planes, J.prm, J.variant, tables, blocks);
        if (rc) return rc;
        for (size_t t = 0; t < tables.size(); ++t) {
            for (int i = 0; i < tables[t].n; ++i) {
                PlanJob q{};
                q.j = tables[t].j[i];
                q.j.first_block += total;
                q.dst = J.packed; q.code = J.planes; q.kind = 0; q.prm_index = k;
                pj.push_back(q);
            }
            total += blocks[t];
        }
        prms[k] = to_params(J.prm);
        PlanJob f{};
        const int f32_total = J.net == 0 ? NetNerf::f32_total : NetNoise::f32_total;
        const int entries = J.net == 0 ? NetNerf::total_entries : NetNoise::total_entries;
        f.j.first_block = total;
        f.dst = reinterpret_cast<float*>(J.packed) + (size_t)entries * nplanes(J.planes) * 256;
        f.code = J.planes; f.kind = J.net == 0 ? 1 : 2; f.prm_index = k;
        pj.push_back(f);
        total += (f32_total + 63) / 64;
    }
    const size_t need = sizeof(PlanHeader) + sizeof(MlpParams) * PLAN_MAX_NETS + sizeof(PlanJob) * pj.size();
    if (need > plan_bytes) return set_error("lush_pack_plan_build: plan buffer too small (lush_pack_plan_bytes)");
    std::vector<char> host(need);
    PlanHeader H{(int)pj.size(), total, n_jobs, 0};
    memcpy(host.data(), &H, sizeof(H));
    memcpy(host.data() + sizeof(H), prms.data(), sizeof(MlpParams) * PLAN_MAX_NETS);
    memcpy(host.data() + sizeof(H) + sizeof(MlpParams) * PLAN_MAX_NETS, pj.data(), sizeof(PlanJob) * pj.size());
    LUSH_HIP(hipMemcpy(plan, host.data(), need, hipMemcpyHostToDevice));       // (set-up call: synchronous, once per model)
    *launch_blocks = total;
    return 0;
}

int lush_pack_plan_run(const void* plan, int launch_blocks, float* zero_buf, long long zero_n, lush_stream_t stream) {
    if (!plan || launch_blocks < 1) return set_error("lush_pack_plan_run: no plan");
    if (zero_n < 0 || (zero_n > 0 && !zero_buf)) return set_error("lush_pack_plan_run: bad zero buffer");
    return launch_pack_plan(plan, launch_blocks, (hipStream_t)stream, zero_n > 0 ? zero_buf : nullptr, zero_n);
}

size_t lush_mlp_stash_bytes(int net, int planes_fwd, int stash_planes, long long P) {
    NetInfo n;
    if (!net_info(net, n) || !code_ok(planes_fwd) || stash_planes < 0 || nplanes(stash_planes) > nplanes(planes_fwd)) return 0;
    // inference on the chain kernel keeps the encoding image in LDS and touches no workspace at all
    if (stash_planes == 0 && mlp_fwd_chain_enabled(planes_fwd)) return 256;
    return stash_layout(n, nplanes(planes_fwd), nplanes(stash_planes), P).total;
}
size_t lush_mlp_dstash_bytes(int net, int planes, long long P) {
    NetInfo n;
    if (!net_info(net, n) || !code_ok(planes)) return 0;
    return dstash_layout(n, nplanes(planes), P).total;
}

int lush_debug_stash_layout(int net, int planes, long long P, long long* o) {
    NetInfo n;
    if (!net_info(net, n)) return set_error("bad net");
    const StashLayout L = stash_layout(n, planes, planes, P);
    o[0] = (long long)L.mask; o[1] = (long long)L.pe;
    for (int l = 0; l < NET_MAX_LAYERS; ++l) o[2 + l] = l < n.NL ? (long long)L.h[l] : -1;
    o[10] = (long long)L.feat; o[11] = (long long)L.hv; o[12] = L.Ppad; o[13] = (long long)L.total;
    o[14] = n.HW; o[15] = n.NL;
    return 0;
}

// a live-point launch: the 8x256 net's one- and two-plane kernels (the 64-points-per-wave kernels of the one-fp16-plane mode and the
// 128-point-tile chain kernels of the bf16-plane modes, with the grouped weight gradients), no older-kernel variant bit; the
// three-plane reference mode keeps its tiled kernels and the backward over all the points
static const int LIVE_OLDER_VARIANTS = LUSH_VARIANT_FWD_HALF | LUSH_VARIANT_FWD_512 | LUSH_VARIANT_BWD_HALF | LUSH_VARIANT_BWD_512 | LUSH_VARIANT_PE_ROWS |
                                       LUSH_VARIANT_HEAD_KERNEL | LUSH_VARIANT_DW_SPLIT;
static bool live_kernels(int net, int planes_f, int planes_b, int variant) {
    return net == 0 && mlp_fwd_chain_enabled(planes_f) && mlp_bwd_chain_enabled(planes_b) && !(variant & LIVE_OLDER_VARIANTS);
}

static int mlp_fwd_impl(int net, int planes, int stash_planes, const float* rays, const float* z, int R, int S,
                        const void* packed, const lush_mlp_params* prm, float* raw, void* stash, int variant, lush_stream_t stream,
                        const int* live_idx, const int* live_cnt) {
    NetInfo n;
    if (!net_info(net, n)) return set_error("lush_mlp_fwd: bad net");
    if (!code_ok(planes)) return set_error("lush_mlp_fwd: planes must be 1..3 or 17 (one fp16 plane)");
    if (stash_planes == PLANES_F16) stash_planes = 1;
    if (stash_planes < 0 || stash_planes > nplanes(planes)) return set_error("lush_mlp_fwd: need 0 <= stash_planes <= planes");
    if (!stash) return set_error("lush_mlp_fwd: stash (or the inference workspace) is required");
    if (R <= 0 || S <= 0) return set_error("lush_mlp_fwd: empty batch");
    const long long P = (long long)R * S;
    // the 64-points-per-wave kernels address d_raw / the point rows by 32-bit byte offsets from a scalar base (16 .. 32 bytes per
    // point): one launch takes at most 2^27 - 1 points (whose stash alone would be 590 GB); split the rays above that
    if (P >= (1LL << 27)) return set_error("lush_mlp_fwd: at most 2^27 - 1 points per launch (split the ray batch)");
    const StashLayout L = stash_layout(n, nplanes(planes), stash_planes, P);
    const bool chain = mlp_fwd_chain_enabled(planes);
    const int mt = chain ? 128 : (planes == PLANES_F16 ? 64 : mlp_fwd_tile(planes));
    MlpFwdArgs a{};
    a.stash_planes = stash_planes;
    a.rays = rays; a.z = z; a.S = S; a.P = (int)P; a.n_tiles = (int)(L.Ppad / mt);
    a.wpk = (const uint4*)packed;
    (void)prm;                                   // biases travel inside `packed` (lush_mlp_pack)
    a.raw = raw;
    a.write_stash = stash_planes > 0;
    char* b = (char*)stash;
    a.mask = (unsigned long long*)(b + L.mask);
    a.mask_dummy = b + L.mask_dummy;
    a.pe = (__bf16*)(b + L.pe);
    a.xd = (float*)(b + L.xd);
    // the weight gradients of the product's one-fp16-plane kernels re-encode gamma(x), gamma(d) from 32 bytes per point; every
    // other kernel / variant reads the 256-byte encoded rows (the caller passes the same variant word to forward and backward)
    a.pe_rows = !(net == 0 && planes == PLANES_F16 && stash_planes == 1 && !(variant & (LUSH_VARIANT_FWD_HALF | LUSH_VARIANT_FWD_512 | LUSH_VARIANT_PE_ROWS)));
    a.h0 = (__bf16*)(b + L.h[0]);
    a.h_stride = n.NL > 1 ? (long long)(L.h[1] - L.h[0]) / 2 : 0;
    a.feat = (__bf16*)(b + L.feat);
    a.hv = (__bf16*)(b + L.hv);
    a.plane_pe = L.Ppad * PE_ROW; a.plane_h = L.Ppad * n.HW; a.plane_hv = L.Ppad * n.HV;
    a.live_idx = live_idx; a.live_cnt = live_cnt;
    if (live_idx || live_cnt) {
        if (!live_idx || !live_cnt) return set_error("lush_mlp_fwd_live: the list and its count come together");
        if (net != 0 || !chain || (variant & LIVE_OLDER_VARIANTS) || stash_planes < 1) return set_error("lush_mlp_fwd_live: the 8x256 net's one- or two-plane kernels with the stash, no older-kernel variant");
    }
    if (chain) return launch_mlp_chain_fwd(net, planes, a, variant, (hipStream_t)stream);
    const int grid = a.n_tiles < 1024 ? a.n_tiles : 1024;
    return launch_mlp_fwd(net, planes, a, grid, (hipStream_t)stream);
}

int lush_mlp_fwd(int net, int planes, int stash_planes, const float* rays, const float* z, int R, int S,
                 const void* packed, const lush_mlp_params* prm, float* raw, void* stash, int variant, lush_stream_t stream) {
    return mlp_fwd_impl(net, planes, stash_planes, rays, z, R, S, packed, prm, raw, stash, variant, stream, nullptr, nullptr);
}
int lush_mlp_fwd_live(int net, int planes, int stash_planes, const float* rays, const float* z, int R, int S, const void* packed,
                      const lush_mlp_params* prm, void* stash, const int* live_idx, const int* live_cnt, int variant, lush_stream_t stream) {
    if (!live_idx || !live_cnt) return set_error("lush_mlp_fwd_live: live_idx and live_cnt are required");
    return mlp_fwd_impl(net, planes, stash_planes, rays, z, R, S, packed, prm, nullptr, stash, variant, stream, live_idx, live_cnt);
}

// prepared != 0: the caller's lush_composite_bwd already computed the loss scale into the dstash header and zeroed the
// feature-factor scratch behind it (lush_march_bwd: no memset, no grad_scale launch here).
static int mlp_bwd_impl(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                        const void* packed_b, const lush_mlp_params* prm, const float* draw, const void* stash,
                        void* dstash, const lush_mlp_grads* g, float* dpts, int variant, lush_stream_t stream, int do_chain,
                        int do_weights, int prepared = 0, const int* live_idx = nullptr, const int* live_cnt = nullptr) {
    NetInfo n;
    if (!net_info(net, n)) return set_error("lush_mlp_bwd: bad net");
    if (live_cnt && !live_kernels(net, planes_f, planes_b, variant)) return set_error("lush_mlp_bwd (live points): the 8x256 net's one- or two-plane kernels, no older-kernel variant");
    const bool x_f16 = planes_f == PLANES_F16;     // the stash was written by the fp16 forward
    if (x_f16) planes_f = 1;
    const bool z_f16 = planes_b == PLANES_F16;     // loss-scaled fp16 gradient chain (one plane)
    const int code_b = planes_b;
    if (z_f16) planes_b = 1;
    if (planes_b < 1 || planes_b > planes_f || planes_f > 3) return set_error("lush_mlp_bwd: need 1 <= planes_b <= planes_f <= 3");
    if (!stash || !dstash) return set_error("lush_mlp_bwd: stash and dstash are required");
    const long long P = (long long)R * S;
    if (R <= 0 || S <= 0 || P >= (1LL << 27)) return set_error("lush_mlp_bwd: 1 .. 2^27 - 1 points per launch (32-bit byte offsets; split the ray batch)");
    const StashLayout L = stash_layout(n, planes_f, planes_f, P);   // gamma rows come last: their plane count does not move the others
    const DStashLayout D = dstash_layout(n, planes_b, P);
    hipStream_t st = (hipStream_t)stream;
    const char* sb = (const char*)stash;
    char* db = (char*)dstash;
    const long long plane_h = L.Ppad * n.HW, plane_hv = L.Ppad * n.HV, plane_pe = L.Ppad * PE_ROW;

    MlpBwdArgs a{};
    const bool chain = mlp_bwd_chain_enabled(code_b);
    if (z_f16 && !chain) return set_error("lush_mlp_bwd: the fp16 gradient chain needs the chain kernels");
    float* gscale = z_f16 ? (float*)(db + D.scale) : nullptr;
    a.scale = gscale;
    a.rays = rays; a.z = z; a.S = S; a.P = (int)P; a.n_tiles = (int)(L.Ppad / (chain ? 128 : mlp_bwd_tile(planes_b)));
    a.wpk = (const uint4*)packed_b;
    a.draw = draw;
    a.mask = (const unsigned long long*)(sb + L.mask);
    __bf16* dzp[NET_MAX_LAYERS];
    for (int l = 0; l < n.NL; ++l) dzp[l] = (__bf16*)(db + D.dz[l]);
    a.dz0 = dzp[0];
    a.dz_stride = n.NL > 1 ? (long long)(D.dz[1] - D.dz[0]) / 2 : 0;
    a.dfeat = (__bf16*)(db + D.dfeat);
    a.dzv = (__bf16*)(db + D.dzv);
    a.plane_h = plane_h; a.plane_hv = plane_hv;
    a.dpts = dpts;
    a.live_idx = live_idx; a.live_cnt = live_cnt;
    const int grid = a.n_tiles < 1024 ? a.n_tiles : 1024;
    int rc = 0;
    bool fac_zeroed = false;       // the feature-factor scratch of the grouped weight gradients was zeroed by the loss-scale launch
    if (do_chain && z_f16 && !prepared) {
        if (!draw) return set_error("lush_mlp_bwd: draw is required");
        // (chain and weight gradients in ONE call -- the noise net's backward: the loss-scale launch also zeroes the scratch the
        // grouped launch accumulates into, instead of a fill launch of its own)
        fac_zeroed = do_weights && planes_b <= 2;
        rc = launch_grad_scale(draw, P * 4, gscale, fac_zeroed ? (float*)(db + D.fac) : nullptr,
                               fac_zeroed ? (long long)((n.HV + DZV_EXT) * (n.HW + 1) + DZV_EXT * (n.HV + 1)) : 0, st);
        if (rc) return rc;
    }
    if (do_chain) rc = chain ? launch_mlp_chain_bwd(net, code_b, a, variant, st) : launch_mlp_bwd(net, planes_b, a, grid, st);
    if (rc || !do_weights) return rc;

    const __bf16* pe = (const __bf16*)(sb + L.pe);
    auto H = [&](int l) { return (const __bf16*)(sb + L.h[l]); };
    const int XV = PE_X_VALID, DV = PE_D_VALID;
    auto dw = [&](const __bf16* Z, long long zplane, int ldz, int n_out, const __bf16* X, long long xplane, int ldx,
                  int xcol0, int k_in, float* dW, int ldw, int wcol0, float* dbias) {
        DwArgs d{};
        d.Z = Z; d.z_plane = zplane; d.ldz = ldz; d.n_out = n_out;
        d.X = X; d.x_plane = xplane; d.ldx = ldx; d.xcol0 = xcol0; d.k_in = k_in;
        d.dW = dW; d.ldw = ldw; d.wcol0 = wcol0; d.db = dbias;
        d.x_f16 = x_f16 ? 1 : 0;
        d.z_f16 = z_f16 ? 1 : 0;
        d.scale = gscale;
        d.Ppad = (int)L.Ppad;
        const int tiles = ((n_out + 127) / 128) * ((k_in + 127) / 128);
        (void)tiles;
        const int splits = dw_splits(L.Ppad);
        long long pps = (L.Ppad + splits - 1) / splits;
        pps = (pps + 31) / 32 * 32;
        d.pts_per_split = (int)pps;
        const int real_splits = (int)((L.Ppad + pps - 1) / pps);
        return launch_dw(planes_b, d, real_splits, st);
    };
    const __bf16* feat = (const __bf16*)(sb + L.feat);
    const __bf16* hv = (const __bf16*)(sb + L.hv);
    if (planes_b <= 2) {
        // one grouped launch: every layer of the pass, the two pairs that share a dZ merged (DwJob::X2)
        DwGroup G{};
        auto job = [&](const __bf16* Z, int ldz, int n_out, const __bf16* X, int ldx, int xcol0, int k_in, float* dW,
                       int ldw, int wcol0, float* dbias) -> DwJob& {
            DwJob& j = G.j[G.n++];
            j.Z = Z; j.ldz = ldz; j.n_out = n_out; j.X = X; j.ldx = ldx; j.xcol0 = xcol0; j.k_in = k_in;
            j.X2 = nullptr; j.ldx2 = 0; j.x2col0 = 0; j.k2_in = 0;
            j.z_plane = (long long)L.Ppad * ldz; j.x_plane = (long long)L.Ppad * ldx; j.x2_plane = 0;
            j.dW = dW; j.ldw = ldw; j.wcol0 = wcol0; j.dW2 = dW; j.ldw2 = ldw; j.wcol2 = 0; j.n_out2 = n_out; j.db = dbias;
            j.pe_mode = 0;
            return j;
        };
        // (the same rule as lush_mlp_fwd: the forward wrote points and view directions instead of encoded rows)
        const bool reencode = net == 0 && x_f16 && planes_b == 1 && !(variant & (LUSH_VARIANT_FWD_HALF | LUSH_VARIANT_FWD_512 | LUSH_VARIANT_PE_ROWS));
        G.xd = reencode ? (const float*)(sb + L.xd) : nullptr;
        auto with_pe = [&](DwJob& j, int col0, int k2, float* dW2, int ldw2, int wcol2) {
            j.X2 = pe; j.ldx2 = PE_ROW; j.x2col0 = col0; j.k2_in = k2; j.x2_plane = plane_pe;
            j.dW2 = dW2; j.ldw2 = ldw2; j.wcol2 = wcol2;
            j.pe_mode = col0 == 0 ? 1 : 2;
        };
        for (int l = 0; l < n.NL; ++l) {
            if (l == 0 && reencode) {      // no rows to stream: the encoding is the job's second input block, computed in the kernel
                DwJob& j = job(dzp[0], n.HW, n.HW, nullptr, PE_ROW, 0, 0, g->w[0], XV, 0, g->b[0]);
                with_pe(j, 0, XV, g->w[0], XV, 0);
            } else if (l == 0) {
                job(dzp[0], n.HW, n.HW, pe, PE_ROW, 0, XV, g->w[0], XV, 0, g->b[0]);
            } else if (l == n.SKIP) {
                DwJob& j = job(dzp[l], n.HW, n.HW, H(l - 1), n.HW, 0, n.HW, g->w[l], XV + n.HW, XV, g->b[l]);
                with_pe(j, 0, XV, g->w[l], XV + n.HW, 0);
            } else {
                DwJob& j = job(dzp[l], n.HW, n.HW, H(l - 1), n.HW, 0, n.HW, g->w[l], n.HW, 0, g->b[l]);
#ifdef LUSH_ABL_H0
                if (l == 1) j.pe_mode = 9;
#else
                (void)j;
#endif
            }
        }
        // feature + views layers: G = dZv^T h_{NL-1} and s = sum dZv into scratch (launch_feat_factor below turns them
        // into dW_feat, db_feat, dW_views[:, :HW], db_views); the gamma(d) columns of dW_views directly
        if (!prm || !prm->w_views || !prm->w_feat || !prm->b_feat) return set_error("lush_mlp_bwd: the grouped weight gradients need the fp32 parameters");
        // One plane: the K<=3 heads ride along (DZV_EXT in lush_mlp.h) unless the caller asks for the separate head kernel.
        const bool fold = planes_b == 1 && !(variant & LUSH_VARIANT_HEAD_KERNEL);
        const bool alpha = net == 0 && g->w_alpha != nullptr && g->b_alpha != nullptr;     // (the noise net's alpha head has no gradient)
        const int ldzv = n.HV + (planes_b == 1 ? DZV_EXT : 0), grow = n.HV + DZV_EXT;
        float* facG = (float*)(db + D.fac);                  // [grow][HW]
        float* facS = facG + (size_t)grow * n.HW;            // [grow]
        float* facH = facS + grow;                           // [DZV_EXT][HV]
        float* facSH = facH + (size_t)DZV_EXT * n.HV;        // [DZV_EXT]
        if (!prepared && !fac_zeroed) LUSH_HIP(hipMemsetAsync(facG, 0, (size_t)(grow * (n.HW + 1) + DZV_EXT * (n.HV + 1)) * 4, st));
        {
            DwJob& j = job(a.dzv, ldzv, fold && alpha ? grow : n.HV, H(n.NL - 1), n.HW, 0, n.HW, facG, n.HW, 0, facS);
            with_pe(j, PE_X, DV, g->w_views, n.HW + DV, n.HW);
            j.n_out2 = n.HV;
        }
        if (fold) job(a.dzv + n.HV, ldzv, DZV_EXT, hv, n.HV, 0, n.HV, facH, n.HV, 0, facSH);   // rgb head: Z = the extra columns, X = views hidden
        // A workgroup ends a job with up to 64 K fp32 atomics on addresses every other slice of the job also adds to: ~50 us per
        // job when 256 slices do it at once, whatever the point count (measured: 32 768 points, 11 jobs in turn, 0.55 ms).  Passes
        // below LUSH_DW_PERJOB_MAX_PTS points therefore run ONE job per workgroup (grid: slices x jobs) with just enough slices to
        // fill the chip once -- one round of atomics in all, 1 / slices of the contenders per address (the same pass: 0.15 ms;
        // 4 096 points 0.107 -> 0.064 ms).  Above it a job's streaming time hides its atomics and every workgroup takes every job of
        // its slice in turn, which balances the narrow jobs.
        // (LUSH_DW_PERJOB_MAX_PTS / _MIN_PTS: lush_mlp.h; a live-point launch makes the same choice on the device)
        int splits = dw_splits(L.Ppad);
        G.per_job = 0;
        G.live_cnt = live_cnt;
        int dev = 0, cus = 256;
        if (current_device_cus(dev, cus) != 0) cus = 256;
        if (L.Ppad <= LUSH_DW_PERJOB_MAX_PTS && !live_cnt) {      // (a live-point launch decides in the kernel: its size is known on the device only)
            long long sp = cus / G.n, most = L.Ppad / LUSH_DW_PERJOB_MIN_PTS;
            if (sp > most) sp = most;
            splits = sp < 1 ? 1 : (int)sp;
            G.per_job = 1;
        }
        long long pps = (L.Ppad + splits - 1) / splits;
        pps = (pps + 31) / 32 * 32;
        G.Ppad = (int)L.Ppad;
        G.pts_per_split = (int)pps;
        G.scale = gscale;
        int grid_x = (int)((L.Ppad + pps - 1) / pps);
        if (!G.per_job && !live_cnt && (variant & LUSH_VARIANT_DW_SPLIT) && cus >= 4 * G.n) {
            // Variant (round 5 experiment, NOT the product's choice: measured slower, see include/lush_march.h): ONE job per
            // workgroup.  A workgroup that walks the ten jobs of its slice drains its ring,
            // flushes 40 K atomics, zeroes and refills the ring ten times -- the launch averaged 86 % of its own streaming rate.
            // Here job j gets n_j of the chip's workgroups, n_j proportional to what a point of that job costs a workgroup, and
            // slices of Ppad / n_j points: every workgroup is busy for the same time and has ONE boundary.  The grid is flat,
            // sum n_j = CUs workgroups: every one is resident from the start.
            // Cost per point, in bytes-equivalent (profiles/r05_dw_jobs.md: per-workgroup durations of a byte-proportional
            // split, -DLUSH_CLOCK): the bytes the job streams (Z row + X row), at least 512 (a narrow job is latency-bound: three
            // small stages in flight), plus DW_PE_COST for a block that is re-encoded from the 32-byte point record (the
            // encoding's ~150 VALU instructions per stage run on four of the eight waves: 27 ns per point).
#ifndef LUSH_DW_PE_COST
#define LUSH_DW_PE_COST 860
#endif
            long long w[DW_MAX_JOBS], W = 0;
            for (int i = 0; i < G.n; ++i) {
                const DwJob& j = G.j[i];
                const bool pe = j.X2 != nullptr && G.xd != nullptr && j.pe_mode != 0;
                w[i] = 2LL * (j.n_out + j.k_in) + (j.X2 && !pe ? 2LL * j.k2_in : 0);
                if (w[i] < 512) w[i] = 512;
                if (pe) w[i] += 32 + LUSH_DW_PE_COST;
                W += w[i];
            }
            int nj[DW_MAX_JOBS], used = 0;
            double frac[DW_MAX_JOBS];
            for (int i = 0; i < G.n; ++i) {
                const double x = (double)cus * (double)w[i] / (double)W;
                nj[i] = (int)x < 1 ? 1 : (int)x;
                frac[i] = x - (int)x;
                used += nj[i];
            }
            while (used < cus) {                       // the CUs left over go to the jobs that were rounded down the most
                int b = 0;
                for (int i = 1; i < G.n; ++i) if (frac[i] > frac[b]) b = i;
                ++nj[b]; frac[b] = -1.0; ++used;
            }
            while (used > cus) {                       // (only when a narrow job was lifted to one workgroup)
                int b = 0;
                for (int i = 1; i < G.n; ++i) if (nj[i] > nj[b]) b = i;
                --nj[b]; --used;
            }
            grid_x = 0;
            for (int i = 0; i < G.n; ++i) {
                long long p = (L.Ppad + nj[i] - 1) / nj[i];
                p = (p + 31) / 32 * 32;
                G.j[i].pps = (int)p;
                G.first[i] = grid_x;
                grid_x += (int)((L.Ppad + p - 1) / p);
            }
            G.first[G.n] = grid_x;
            G.per_job = 2;
        }
        rc = launch_dw_group(G, grid_x, planes_b, x_f16, z_f16, st);
        if (rc) return rc;
        FeatFactorArgs F{};
        F.G = facG; F.s = facS;
        if (fold) {
            F.Hd = facH; F.sH = facSH; F.g_w_rgb = g->w_rgb; F.g_b_rgb = g->b_rgb;
            F.g_w_alpha = alpha ? g->w_alpha : nullptr; F.g_b_alpha = alpha ? g->b_alpha : nullptr;
        }
        F.w_views = prm->w_views; F.w_feat = prm->w_feat; F.b_feat = prm->b_feat;
        F.g_w_feat = g->w_feat; F.g_b_feat = g->b_feat; F.g_w_views = g->w_views; F.g_b_views = g->b_views;
        F.HW = n.HW; F.HV = n.HV; F.ldv = n.HW + DV;
        rc = launch_feat_factor(F, st);
        if (rc || fold) return rc;
        return launch_head_dw(planes_b, x_f16, draw, P, hv, plane_hv, n.HV, H(n.NL - 1), plane_h, n.HW, g->w_rgb, g->b_rgb,
                              net == 0 ? g->w_alpha : nullptr, net == 0 ? g->b_alpha : nullptr, st, live_cnt);
    }
    // three planes (test reference): one launch per layer
    for (int l = 0; l < n.NL && !rc; ++l) {
        const __bf16* Z = dzp[l];
        if (l == 0) {
            rc = dw(Z, plane_h, n.HW, n.HW, pe, plane_pe, PE_ROW, 0, XV, g->w[0], XV, 0, g->b[0]);
        } else if (l == n.SKIP) {
            rc = dw(Z, plane_h, n.HW, n.HW, pe, plane_pe, PE_ROW, 0, XV, g->w[l], XV + n.HW, 0, g->b[l]);
            if (!rc) rc = dw(Z, plane_h, n.HW, n.HW, H(l - 1), plane_h, n.HW, 0, n.HW, g->w[l], XV + n.HW, XV, nullptr);
        } else {
            rc = dw(Z, plane_h, n.HW, n.HW, H(l - 1), plane_h, n.HW, 0, n.HW, g->w[l], n.HW, 0, g->b[l]);
        }
    }
    if (rc) return rc;
    rc = dw(a.dfeat, plane_h, n.HW, n.HW, H(n.NL - 1), plane_h, n.HW, 0, n.HW, g->w_feat, n.HW, 0, g->b_feat);
    if (rc) return rc;
    rc = dw(a.dzv, plane_hv, n.HV, n.HV, feat, plane_h, n.HW, 0, n.HW, g->w_views, n.HW + DV, 0, g->b_views);
    if (rc) return rc;
    rc = dw(a.dzv, plane_hv, n.HV, n.HV, pe, plane_pe, PE_ROW, PE_X, DV, g->w_views, n.HW + DV, n.HW, nullptr);
    if (rc) return rc;
    return launch_head_dw(planes_b, x_f16, draw, P, hv, plane_hv, n.HV, H(n.NL - 1), plane_h, n.HW, g->w_rgb, g->b_rgb,
                          net == 0 ? g->w_alpha : nullptr, net == 0 ? g->b_alpha : nullptr, st);
}

int lush_mlp_bwd(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                 const void* packed_b, const lush_mlp_params* prm, const float* draw, const void* stash,
                 void* dstash, const lush_mlp_grads* g, float* dpts, int variant, lush_stream_t stream) {
    return mlp_bwd_impl(net, planes_f, planes_b, rays, z, R, S, packed_b, prm, draw, stash, dstash, g, dpts, variant, stream, 1, 1);
}
int lush_mlp_bwd_chain(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                       const void* packed_b, const lush_mlp_params* prm, const float* draw, const void* stash,
                       void* dstash, float* dpts, int variant, lush_stream_t stream) {
    return mlp_bwd_impl(net, planes_f, planes_b, rays, z, R, S, packed_b, prm, draw, stash, dstash, nullptr, dpts, variant, stream, 1, 0);
}
int lush_mlp_bwd_weights(int net, int planes_f, int planes_b, int R, int S, const lush_mlp_params* prm, const float* draw,
                         const void* stash, void* dstash, const lush_mlp_grads* g, int variant, lush_stream_t stream) {
    return mlp_bwd_impl(net, planes_f, planes_b, nullptr, nullptr, R, S, nullptr, prm, draw, stash, dstash, g, nullptr, variant, stream, 0, 1);
}

int lush_mlp_bwd_chain_live(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                            const void* packed_b, const lush_mlp_params* prm, const float* draw_c, const void* stash, void* dstash,
                            float* dpts, const int* live_idx, const int* live_cnt, int variant, lush_stream_t stream) {
    if (!live_idx || !live_cnt) return set_error("lush_mlp_bwd_chain_live: live_idx and live_cnt are required");
    // (the loss scale is taken over all R*S rows of draw_c: lush_live_compact zeroed the rows behind the list)
    return mlp_bwd_impl(net, planes_f, planes_b, rays, z, R, S, packed_b, prm, draw_c, stash, dstash, nullptr, dpts, variant, stream, 1, 0, 0,
                        live_idx, live_cnt);
}
int lush_mlp_bwd_weights_live(int net, int planes_f, int planes_b, int R, int S, const lush_mlp_params* prm, const float* draw_c,
                              const void* stash, void* dstash, const lush_mlp_grads* g, const int* live_cnt, int variant,
                              lush_stream_t stream) {
    if (!live_cnt) return set_error("lush_mlp_bwd_weights_live: live_cnt is required");
    return mlp_bwd_impl(net, planes_f, planes_b, nullptr, nullptr, R, S, nullptr, prm, draw_c, stash, dstash, g, nullptr, variant, stream, 0, 1, 0,
                        nullptr, live_cnt);
}

}  // extern "C"

// ---- for lush_march_bwd (lush_march_abi.hip), whose compositing backward has already prepared the dstash header ----
namespace lush {
int mlp_bwd_chain_prepared(int net, int planes_f, int planes_b, const float* rays, const float* z, int R, int S,
                           const void* packed_b, const void* prm, const float* draw, const void* stash,
                           void* dstash, float* dpts, int variant, void* stream, const int* live_idx, const int* live_cnt) {
    return mlp_bwd_impl(net, planes_f, planes_b, rays, z, R, S, packed_b, (const lush_mlp_params*)prm, draw, stash, dstash, nullptr, dpts,
                        variant, (lush_stream_t)stream, 1, 0, 1, live_idx, live_cnt);
}
int mlp_bwd_weights_prepared(int net, int planes_f, int planes_b, int R, int S, const void* prm, const float* draw,
                             const void* stash, void* dstash, const void* g, int variant, void* stream, const int* live_cnt) {
    return mlp_bwd_impl(net, planes_f, planes_b, nullptr, nullptr, R, S, nullptr, (const lush_mlp_params*)prm, draw, stash, dstash,
                        (const lush_mlp_grads*)g, nullptr, variant, (lush_stream_t)stream, 0, 1, 1, nullptr, live_cnt);
}
bool mlp_live_kernels(int net, int planes_f, int planes_b, int variant) { return live_kernels(net, planes_f, planes_b, variant); }
// where the header of a dstash holds {scale, 1/scale, work, work} and the scratch the weight-gradient launch accumulates into
bool mlp_dstash_header(int net, int planes_b, long long P, void* dstash, float** scale4, float** zero_buf, long long* zero_n) {
    NetInfo n;
    if (!net_info(net, n) || !dstash) return false;
    const int ns = planes_b == PLANES_F16 ? 1 : planes_b;
    const DStashLayout D = dstash_layout(n, ns, P);
    char* db = (char*)dstash;
    *scale4 = planes_b == PLANES_F16 ? (float*)(db + D.scale) : nullptr;
    const bool fac = ns <= 2;
    *zero_buf = fac ? (float*)(db + D.fac) : nullptr;
    *zero_n = fac ? (long long)((n.HV + DZV_EXT) * (n.HW + 1) + DZV_EXT * (n.HV + 1)) : 0;
    return true;
}
}  // namespace lush

