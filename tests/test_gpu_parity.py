"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel and the end-to-end
path, through the C ABI, against the CPU oracle and the reference-made golden fixtures.

Tolerances (normalised max error = max|a-b| / max|b|):
  * forward outputs (rgb_map, rgb0, acc_map, tone-mapped colours, loss): 1e-4  -- the
    north-star bound; depth_map / z_std 1e-3 / 2e-3 because sample_pdf amplifies 1-ulp cdf
    differences by 1/pdf (a property of the reference algorithm in fp32);
  * kernel-level backward vs torch autograd with the GPU's own ReLU decisions: 2e-5 (3 planes),
    2e-4 (2 planes), 3e-2 (1 plane = plain bf16);
  * END-TO-END gradients, primary gate: the oracle's forward_train with every MLP ReLU replaced by the decision the
    GPU took (read back from its stash), torch autograd on that -- per parameter tensor 2e-4 in the (2,2) mode, the
    measured bound gpu_diag.MASKED_GATE in the others; plus the count of decisions that differ from the fp32 oracle;
  * secondary: end-to-end gradients vs the reference fixture, un-masked: 3e-2 (includes the discrete effect of the
    few flipped ReLU kinks);
  * a 40-step training trajectory of the reference itself (tests/golden/train_trajectory.npz) that every precision
    mode must follow inside a stated band.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SECTIONS = ["t_zgrid_pack", "t_gen_rays", "t_composite", "t_sample", "t_mlp_fwd", "t_mlp_ragged", "t_mlp_bwd", "t_rbk", "t_warp_ndc", "t_mix", "t_blur_mix",
            "t_march_e2e", "t_train_e2e", "t_lindisp_white", "t_eval_forward", "t_consistency", "t_faults", "t_draws", "t_train_c1", "t_consist_step"]


@pytest.fixture(scope="module")
def diag():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the HIP path has no CPU fallback")
    from lush_nerf_amd import lib
    lib.load()
    from tests import gpu_diag
    return gpu_diag


@pytest.mark.parametrize("section", SECTIONS)
def test_kernel_parity(diag, section):
    diag.RESULTS.clear()
    getattr(diag, section)()
    torch.cuda.synchronize()
    bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
    assert diag.RESULTS, "section produced no checks"
    assert not bad, f"{len(bad)} of {len(diag.RESULTS)} checks failed: {bad[:8]}"


def test_fp16_forward_mode(diag, monkeypatch):
    """Optional mode (h,1): ONE fp16 plane in the forward.  Render outputs stay inside the 1e-4 bound
    (measured 2.7e-5) at a third of the MFMA work; the price is that more ReLU decisions differ from the fp32
    oracle's (counted and gated in t_train_e2e) and a looser z_std (sample_pdf conditioning), which is why it is
    not the bench headline.  Gradient gates are gpu_diag's mode-aware ones (masked primary, fixture secondary)."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes("h,1"))
    diag.RESULTS.clear()
    diag.t_march_e2e()
    res = {n: e for n, e, t, ok in diag.RESULTS}
    for n, e in res.items():
        if n.endswith("rgb_map") or n.endswith("rgb0") or n.endswith("noise_rgb") or n.endswith("acc_map"):
            assert e < 1e-4, (n, e)
        if n.endswith("depth_map"):
            assert e < 1e-3, (n, e)
        if n.endswith("z_std (worst ray)"):
            assert e < 5e-2, (n, e)
    diag.RESULTS.clear()
    diag.t_train_e2e()
    bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
    assert not bad, bad[:8]


def test_outputs_stay_inside_the_bound_on_a_trained_like_field(diag):
    """The render outputs of the headline mode on the density field bench.py's `trained_like` workload starts from
    (synth.all_weights(trained_like=True): sigma = 3000 w.h + 20 -- empty space and surfaces, a x3000 density head that amplifies the
    forward's operand rounding) and on one twice as sharp: the blurred and the sharp colours of both passes within the north star's 1e-4
    of the fp32 oracle in (h,h) (measured 4.2e-5 / 5.3e-5: the initialisation's 1.2e-5 grows with the sharpness, the bound holds),
    2e-5 in the strict mode -- where the fp32 oracle itself is 3e-6 / 2e-5 from float64 (tools/sharpness_parity.py prints the table)."""
    import argparse
    from lush_nerf_amd import model as M, ops, synth
    from oracle import lush_oracle as O
    dev = torch.device("cuda:0")
    H, W, F, n_img, n, Ns, Ni = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 30, 64, 64, 64
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=Ni, netdepth=8,
                              netwidth=256, netdepth_fine=8, netwidth_fine=256, rgb_activate="sigmoid", sigma_activate="relu",
                              tone_mapping_type="gamma", render_rmnearplane=80)
    b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(n, 1000, n_img).items()}
    d = {k: torch.from_numpy(v) for k, v in synth.draws(n * 5, Ns, Ni, 0).items()}
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    for tl in (True, (6000.0, 40.0)):
        w = synth.all_weights(n_img, 0, rbk_scale=2.0e4, trained_like=tl)
        p = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
        with torch.no_grad():
            ref = O.forward_train(p, H, W, F, b["rays"], b["images_idx"], Ns, Ni, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws=d)
        for name, prec, tol in (("h,h", ops.Precision(ops.PLANES_F16, ops.PLANES_F16), 1e-4), ("2,2", ops.Precision(2, 2), 2e-5)):
            net = M.NeRFAll(args, M.RBK(n_img, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4), precision=prec)
            M.load_reference_weights(net, w)
            net = net.to(dev).train()
            with torch.no_grad():
                out = net(H, W, K, chunk=1 << 20, rays=b["rays"].to(dev), rays_info={"images_idx": b["images_idx"].to(dev)}, retraw=True,
                          force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"].to(dev), perturb=1., N_importance=Ni, N_samples=Ns,
                          use_viewdirs=True, white_bkgd=False, raw_noise_std=1., inference=False, near=0., far=1.,
                          draws={k: v.to(dev) for k, v in d.items()})
            for oname, i in (("rgb_blur", 0), ("rgb0_blur", 1), ("rgb", 5), ("rgb0", 6)):
                err = float((out[i].cpu() - ref[i]).abs().max() / ref[i].abs().max())
                assert err < tol, (tl, name, oname, err)
            assert net.read_faults() == 0


@pytest.mark.parametrize("planes", ["2,1", "2,h", "h,h"])
def test_headline_mode_end_to_end(diag, planes, monkeypatch):
    """The faster precision modes -- (h,h) is the bench headline, (2,1) was round 1's, (2,h) is the fall-back -- on the
    reference fixtures: forward within 1e-4, gradients through gpu_diag's mode-aware gates (masked oracle primary,
    fixture secondary)."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes(planes))
    for section in ("t_march_e2e", "t_train_e2e", "t_consistency", "t_lindisp_white"):
        diag.RESULTS.clear()
        getattr(diag, section)()
        bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
        assert not bad, bad[:8]


@pytest.mark.parametrize("section", ["t_train_c1", "t_consist_step"])
def test_unverified_gradient_paths_in_the_headline_mode(diag, section, monkeypatch):
    """Round 5: the two steps whose backward had never met the oracle, in the bench's mode (h,h) (the fp32-equivalent mode runs
    them in test_kernel_parity).  t_train_c1 = BASELINE config 1's step exactly as bench.py times it (Trainer.step_coarse_only:
    render_infer at N_rand 256, 32 + 0, one summed loss gradient, Adam), eagerly AND as the replayed HIP graph, against the
    reference fixture `train_c1` and the masked float64 oracle (models/lushnerf.py:679-763, :354-479).  t_consist_step =
    Trainer.step(batch, i, consist=...) around noisenerf_start_iter (run_lushnerf.py:625-661) against the reference fixture
    `train_consist`, incl. the `>` / `>=` edge and step_graph's fall-back."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes("h,h"))
    diag.RESULTS.clear()
    getattr(diag, section)()
    torch.cuda.synchronize()
    bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
    assert diag.RESULTS and not bad, bad[:8]


def test_plain_bf16_is_outside_the_parity_bound(diag, monkeypatch):
    """Documents WHY the headline is not plain bf16: (1,1) misses the 1e-4 forward bound."""
    monkeypatch.setattr(diag, "E2E_PLANES", (1, 1))
    diag.RESULTS.clear()
    diag.t_march_e2e()
    errs = [e for n, e, t, ok in diag.RESULTS if n.endswith("rgb_map")]
    assert errs and max(errs) > 1e-4 and max(errs) < 1e-2


def _model(Ni=64, precision=(2, 2), seed=0):
    import argparse
    from lush_nerf_amd import model as M, ops, synth
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)
    net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                    precision=ops.Precision(*precision))
    M.load_reference_weights(net, synth.all_weights(30, seed, sharp=True))
    return net.to("cuda:0")


# Render-output bounds per forward mode at full size: 1e-4 is the north-star bound for both; depth_map as everywhere 1e-3.
FULL_SIZE_MODES = ["2,2", "h,h"]


@pytest.mark.parametrize("planes", FULL_SIZE_MODES)
def test_full_size_properties(diag, planes):
    """BASELINE config 2 size (20 480 marched rays, 64+64): size-independent properties and 64 rays against the oracle,
    in the fp32-equivalent mode AND in the bench headline mode (h,h): 20 480 fine tiles = 40 per persistent workgroup of
    mlp_chain_fwd_half_kernel (80 for the 512-register kernel), i.e. the weight stream wraps into the next tile as it
    does in bench.py -- both the inference variant (no stash) and the stash-writing variant the training step runs."""
    from lush_nerf_amd import ops, synth
    from oracle import lush_oracle as O
    dev = torch.device("cuda:0")
    prec = ops.Precision(*ops.parse_planes(planes))
    net = _model(precision=(prec.fwd, prec.bwd)).train()
    R, Ns, Ni = 20480, 64, 64
    b = synth.ray_batch(R, 5)
    batch = ops.PackRays.apply(torch.from_numpy(b["rays"]).to(dev), synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF,
                               True, 0., 1.)
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(R, Ns, Ni, 5).items()}
    kw = dict(N_samples=Ns, N_importance=Ni, perturb=1., raw_noise_std=1., retraw=True)
    with torch.no_grad():
        cfg = ops.MarchCfg(Ns, Ni, 1., 1., precision=prec, want_grad=False)
        coarse, fine = net.mlp_coarse.tensors(), net.mlp_fine.tensors()
        out = ops.March.apply(batch, cfg, d, len(coarse), *coarse, *fine)
        out2 = ops.March.apply(batch, cfg, d, len(coarse), *coarse, *fine)
    rgb, depth, acc, density, raw, weights, z = out[:7]
    assert all(torch.equal(a, b_) for a, b_ in zip(out, out2)), "forward is not deterministic"
    # the stash-writing variant of the same kernels (what a training step launches) gives the same render outputs
    cfg_g = ops.MarchCfg(Ns, Ni, 1., 1., precision=prec, want_grad=True)
    out_g = ops.March.apply(batch, cfg_g, d, len(coarse), *coarse, *fine)
    assert out_g[0].requires_grad, "the training variant did not run"
    for i in (0, 1, 2, 7):
        assert diag.util.relerr(out_g[i].detach(), out[i]) < 1e-6, (i, diag.util.relerr(out_g[i].detach(), out[i]))
    del out_g
    assert bool((z[:, 1:] >= z[:, :-1]).all()), "merged z_vals not sorted"
    # the last sample has alpha == 1, so the weights telescope to exactly 1 (models/lushnerf.py:338-341)
    assert float((acc - 1).abs().max()) < 2e-6
    assert float((weights.sum(-1) - acc).abs().max()) < 1e-6
    assert float(rgb.min()) >= 0 and float(rgb.max()) <= 1 + 1e-6
    assert float(depth.min()) >= 0 and float(depth.max()) <= 1 + 1e-6
    assert bool((density >= 0).all()) and bool(torch.isfinite(raw).all())
    # chunk invariance: rays are independent, so chunked rendering is bit-identical (models/lushnerf.py:800)
    sl = slice(4096, 8192)
    with torch.no_grad():
        part = ops.March.apply(batch[sl].contiguous(), cfg, {k: v[sl].contiguous() for k, v in d.items()},
                               len(coarse), *coarse, *fine)
    assert torch.equal(part[0], rgb[sl]) and torch.equal(part[6], z[sl])
    # spot parity at full size: 64 random rays against the oracle
    idx = torch.arange(0, R, R // 64)[:64]
    p = {k: v.detach().cpu() for k, v in _canon(net).items()}
    with torch.no_grad():
        ref = O.render_rays(p, batch[idx].cpu(), Ns, perturb=1., N_importance=Ni, raw_noise_std=1.,
                            draws={k: v[idx].cpu() for k, v in d.items()}, with_noise_branch=False)
    e_rgb, e_rgb0, e_depth = (diag.util.relerr(rgb[idx], ref["rgb_map"]), diag.util.relerr(out[7][idx], ref["rgb0"]),
                              diag.util.relerr(depth[idx], ref["depth_map"]))
    print(f"full size {planes}: rgb_map {e_rgb:.2e}, rgb0 {e_rgb0:.2e}, depth_map {e_depth:.2e} (64 rays against the oracle)")
    assert e_rgb < 1e-4 and e_rgb0 < 1e-4 and e_depth < 1e-3


@pytest.mark.parametrize("sampling", [(512, 64, 64), (256, 128, 128)], ids=["64+64", "128+128"])
@pytest.mark.parametrize("planes", ["2,2", "h,h"])
def test_bench_regime_against_the_oracle(diag, planes, sampling, monkeypatch):
    """N_rand 512, blur kernel on (2 560 marched rays, 327 680 fine points = 2 560 tiles): every persistent workgroup of
    the forward / backward chain kernels walks >= 5 tiles (10 for the 512-register kernels) and dw_group_kernel many
    slices -- the regime of the bench -- with outputs at 1e-4 against the fp32 oracle and per-tensor gradients against
    the masked float64 oracle (gpu_diag.masked_grad_check with the existing floors).  The second sampling is BASELINE
    config 5's (128 + 128) at the same number of fine points (N_rand 256): the S = 256 compositing kernels, the ray
    reduction over 256 samples and the chain / weight-gradient kernels in the headline mode, forward AND backward
    (models/lushnerf.py:296-352, 481-583)."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes(planes))
    diag.RESULTS.clear()
    n, Ns, Ni = sampling
    diag.t_train_bench_regime(n=n, Ns=Ns, Ni=Ni)
    bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
    assert diag.RESULTS and not bad, bad[:8]


@pytest.mark.parametrize("planes", ["h,h", "2,h"])
def test_trained_like_regime_against_the_oracle(diag, planes, monkeypatch):
    """Round 6: a training step from the density field bench.py's `trained_like` workload starts from (empty space and surfaces behind a
    x3000 density head; N_rand 128 with the blur kernel on = 81 920 fine points) in the headline mode and in the mode a user takes
    for accurate gradients: outputs at 1e-4 of the fp32 oracle (measured 8.5e-5 / 5.9e-5: sample_pdf's knot discontinuities show at
    this sharpness in every mode), every gradient tensor against the float64 oracle run with the GPU's ReLU decisions.  The gates are
    that regime's own (gpu_diag COS_GATE_BY_TAG / WELL_FLOOR_BY_TAG / MASKED_CAP_BY_TAG, 2 - 3 x what was measured): the derivative
    with respect to the ray geometry through a sharp field is a cancelling sum on which the fp32 oracle itself is 0.6 - 1.5 % from
    float64; (h,h) sits at 10 - 21 % on the blur-kernel network's tensors and d(rays) (1 - cos 1.2e-2), 7e-3 (median) on the 8 x 256
    networks'; (2,h) at 1 - 1.7 % and 6e-4 (profiles/r06_trained_like_grads.md)."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes(planes))
    diag.RESULTS.clear()
    diag.t_train_bench_regime(n=128, seed=0, trained_like=True, min_tiles=512)
    bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
    assert diag.RESULTS and not bad, bad[:8]


def _canon(net):
    out = {}
    for k, v in net.state_dict().items():
        if k.startswith("blur_kernel_net.RBK."):
            k = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        elif k.startswith("blur_kernel_net.view_embed_layer.") or k.startswith("dbk_view_embedding."):
            k = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        out[k] = v
    return out


def test_trainer_step_matches_torch_adam(diag):
    """One Trainer.step == reference semantics: same loss, params move as torch.optim.Adam moves them,
    parameters the reference leaves with grad=None are untouched."""
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    net = _model(seed=3)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=0,
                 allkernel_start_iter=1 << 30)
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(64, 9).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(64 * 5, 64, 64, 9).items()}
    loss = tr.step(b, 0, draws=d)
    assert bool(torch.isfinite(loss))
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    for k, v in net.named_parameters():
        if "mlp_noise_coarse.alpha_linear" in k:
            assert torch.equal(v.detach(), before[k]), k          # never stepped
            continue
        ref = before[k].clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=5e-4)
        ref.grad = grads[k].clone()
        opt.step()
        assert diag.util.relerr(v.detach(), ref.detach()) < 1e-6, k
    # naive phase: RBK and noise MLP are not stepped
    net2 = _model(seed=3)
    tr2 = Trainer(net2, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, kernel_start_iter=10)
    b4 = {k: v.detach().clone() for k, v in net2.named_parameters()}
    tr2.step(b, 0, draws={k: v[:64] for k, v in d.items()})
    for k, v in net2.named_parameters():
        moved = not torch.equal(v.detach(), b4[k])
        frozen = k.startswith("blur_kernel_net") or k.startswith("mlp_noise_coarse") or k.startswith("mlp_rbk") \
            or k.startswith("dbk_view_embedding")
        assert moved != frozen, k


def test_micro_batched_step_equals_full_step(diag):
    """Trainer(micro_batch=k) accumulates per-slice gradients: same loss and gradients as one big step."""
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(96, 4).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(96 * 5, 64, 64, 4).items()}
    res = []
    for mb in (0, 32):
        net = _model(seed=6, precision=(2, 2))
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, micro_batch=mb)
        loss = tr.step(b, 0, draws=d)
        res.append((float(loss), tr.flat.grad.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-6 * max(1.0, abs(res[0][0]))
    assert diag.util.relerr(res[1][1], res[0][1]) < 2e-4      # fp32 atomics: summation order differs


@pytest.mark.parametrize("cfg", ["C3", "C5"])
def test_large_configs_step(diag, cfg):
    """BASELINE configs 3 (8192 rays, 64+64) and 5 (16 384 rays, 128+128, the HBM stress case) in the mode bench.py
    reports them in, (h,h): one optimisation step's forward + backward with bounded memory (micro-batches of 4096 input
    rays), no numerical fault, and the gradient of the step equal to the gradient of the same step cut into micro-batches
    of 2048 -- the loss is a mean over rays, so the two differ only in the order of the fp32 sums and in the per-launch
    power-of-two loss scale (2e-4, as for config 2 in test_full_size_backward_is_additive_over_rays).  Together with
    test_bench_regime_against_the_oracle[128+128] (the same kernels against the float64 oracle at 327 680 points) this is
    the gradient-side evidence for the C3 / C5 bench lines."""
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    import bench
    dev = torch.device("cuda:0")
    n, Ns, Ni = (8192, 64, 64) if cfg == "C3" else (16384, 128, 128)
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 8).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, 8).items()}
    res = []
    for mb in (4096, 2048):
        net = bench.make_model(bench.model_args(Ni), dev, diag.ops.Precision(*diag.ops.parse_planes("h,h")))
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, micro_batch=mb)
        torch.cuda.reset_peak_memory_stats()
        loss = tr.step(b, 0, draws=d)
        torch.cuda.synchronize()
        assert tr.faults() == 0
        assert bool(torch.isfinite(loss)) and bool(torch.isfinite(tr.flat.param).all())
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        print(f"{cfg} micro-batch {mb}: loss {float(loss):.6f}, peak torch memory {peak:.1f} GiB")
        assert peak < 200
        res.append((float(loss), tr.flat.grad.clone()))
        del tr, net
    assert float(res[0][1].abs().max()) > 0
    assert abs(res[0][0] - res[1][0]) < 2e-6 * max(1.0, abs(res[0][0]))
    e_all = diag.util.relerr(res[1][1], res[0][1])
    e_mlp = diag.util.relerr(res[1][1][:2 * 595844], res[0][1][:2 * 595844])
    print(f"{cfg} additivity (h,h): whole gradient {e_all:.2e}, MLP segment {e_mlp:.2e}")
    assert e_all < 2e-4 and e_mlp < 2e-4


def test_eval_path_runs(diag):
    """render_path / eval forward: small image, no grad, finite outputs."""
    from lush_nerf_amd import synth
    dev = torch.device("cuda:0")
    net = _model().eval()
    H, W, F = 24, 40, 35.0
    K = [[F, 0, W / 2], [0, F, H / 2], [0, 0, 1]]
    poses = torch.from_numpy(synth.poses(2, 1)).to(dev)
    rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
              inference=True, near=0., far=1.)
    rgbs, noise, depths = net(H, W, K, chunk=512, poses=poses, render_kwargs=rk)
    assert rgbs.shape == (2, H, W, 3) and noise.shape == (2, H, W, 3) and depths.shape == (2, H, W)
    assert bool(torch.isfinite(rgbs).all()) and bool(torch.isfinite(noise).all())


@pytest.mark.parametrize("planes,tol", [("2,2", 2e-4), ("h,h", 2e-4)])
def test_full_size_backward_is_additive_over_rays(diag, planes, tol):
    """BASELINE config 2 size, backward: the loss is a mean over rays, so the gradient of the full 4096-ray batch
    equals the sum of the gradients of its two halves (each weighted by its share), 2e-4 in the fp32-equivalent mode and
    in the headline mode (h,h) alike: the halves run the same 16-bit arithmetic per point, their loss scales are powers
    of two, and only the order of the fp32 sums differs (measured 4.6e-7 / 4.0e-7)."""
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    n = 4096
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 4).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, 64, 64, 4).items()}
    res = []
    for mb in (0, 2048):
        net = _model(seed=6, precision=diag.ops.parse_planes(planes))
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, micro_batch=mb)
        loss = tr.step(b, 0, draws=d)
        res.append((float(loss), tr.flat.grad.clone()))
        assert tr.faults() == 0
        del tr, net
    assert abs(res[0][0] - res[1][0]) < 2e-6 * max(1.0, abs(res[0][0]))
    e_all = diag.util.relerr(res[1][1], res[0][1])
    a, bb = 0, 2 * 595844          # per segment too: coarse + fine MLP block
    e_mlp = diag.util.relerr(res[1][1][a:bb], res[0][1][a:bb])
    print(f"additivity {planes}: whole gradient {e_all:.2e}, MLP segment {e_mlp:.2e} (tol {tol:.0e})")
    assert e_all < tol and e_mlp < tol


@pytest.mark.parametrize("planes", FULL_SIZE_MODES)
@pytest.mark.parametrize("cfg", ["C3", "C5"])
def test_large_configs_spot_parity(diag, cfg, planes):
    """BASELINE configs 3 / 5 at their full marched-ray counts (40 960 rays 64+64; 81 920 rays 128+128), forward:
    64 rays spread over the batch against the oracle at 1e-4, as for config 2, in the fp32-equivalent and the headline mode."""
    from lush_nerf_amd import ops, synth
    from oracle import lush_oracle as O
    dev = torch.device("cuda:0")
    R, Ns, Ni = (40960, 64, 64) if cfg == "C3" else (81920, 128, 128)
    prec = ops.Precision(*ops.parse_planes(planes))
    net = _model(Ni=Ni, precision=(prec.fwd, prec.bwd)).train()
    b = synth.ray_batch(R, 15)
    batch = ops.PackRays.apply(torch.from_numpy(b["rays"]).to(dev), synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, True, 0., 1.)
    idx = torch.arange(0, R, R // 64)[:64]
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(R, Ns, Ni, 15).items()}
    with torch.no_grad():
        cfgm = ops.MarchCfg(Ns, Ni, 1., 1., precision=prec, want_grad=False)
        coarse, fine = net.mlp_coarse.tensors(), net.mlp_fine.tensors()
        out = ops.March.apply(batch, cfgm, d, len(coarse), *coarse, *fine)
        p = {k: v.detach().cpu() for k, v in _canon(net).items()}
        ref = O.render_rays(p, batch[idx].cpu(), Ns, perturb=1., N_importance=Ni, raw_noise_std=1.,
                            draws={k: v[idx].cpu() for k, v in d.items()}, with_noise_branch=False)
    assert diag.util.relerr(out[0][idx], ref["rgb_map"]) < 1e-4
    assert diag.util.relerr(out[7][idx], ref["rgb0"]) < 1e-4
    assert diag.util.relerr(out[1][idx], ref["depth_map"]) < 1e-3
    z = out[6]
    assert bool((z[:, 1:] >= z[:, :-1]).all()) and float((out[2] - 1).abs().max()) < 4e-6


# Band of the training-trajectory test: max over the 40 steps of |loss_gpu - loss_reference| / loss_reference.
# The reference run is fp32 torch on CPU; rays and draws are fresh every step, so the curve tests the composed
# forward + backward + Adam + lr path, and errors compound through the parameters.
# Measured (round 2, final kernels, 6 repetitions x 5 modes on one box): typically 2e-4 .. 8e-4 in EVERY mode, the
# fp32-equivalent (2,2) included, with outliers of 1.2e-3 ((2,2)) and 1.8e-3 ((2,1)) -- the repetitions differ only in the
# order of the fp32 atomics of the weight-gradient sums, and what accumulates over the steps is the chaotic sensitivity
# of the run itself (flipped ReLU kinks, Adam's m / sqrt(v) in its first steps), not the arithmetic of a mode.
# One band for all: twice the largest deviation seen in those 30 runs.
TRAJ_BAND = {"2,2": 4e-3, "2,1": 4e-3, "h,1": 4e-3, "2,h": 4e-3, "h,h": 4e-3}


@pytest.mark.parametrize("planes", ["2,2", "2,1", "h,1", "2,h", "h,h"])
def test_training_trajectory_follows_the_reference(diag, planes):
    """40 optimisation steps (32 rays x 5 motions, 64+64, blur kernel on, fresh rays/draws each step) of the REAL
    reference (make_golden.case_trajectory: its model, torch.optim.Adam in its two-group set-up, its lr rule) against
    Trainer.step in each precision mode.  This, not a fixed gradient tolerance, is what qualifies a mode as headline."""
    import numpy as np
    from lush_nerf_amd import model as M, ops, synth
    from lush_nerf_amd.trainer import Trainer
    g = diag.util.golden("train_trajectory")
    n, Ns, Ni, seed, steps = (int(x) for x in g["meta"])
    dev = torch.device("cuda:0")
    import argparse
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
    net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                    precision=ops.Precision(*ops.parse_planes(planes)))
    M.load_reference_weights(net, synth.all_weights(30, seed, sharp=True, rbk_scale=2.0e4))
    net = net.to(dev)
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, kernel_start_iter=0, allkernel_start_iter=0)
    losses = []
    for s in range(steps):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, seed, 30, step=s).items()}
        d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, seed, step=s).items()}
        losses.append(float(tr.step(b, s, draws=d)))
    assert tr.faults() == 0
    ref = np.asarray(g["losses"], dtype=np.float64)
    dev_rel = np.abs(np.asarray(losses) - ref) / ref
    sd = dict(_canon(net))
    sd.update(net.state_dict())                      # the fixture lists some aliases under their raw names
    keys = [str(k) for k in g["final_keys"]]
    norms = np.array([float(sd[k].double().norm()) for k in keys])
    dn = np.abs(norms - g["final_norms"]) / np.maximum(g["final_norms"], 1e-12)
    dw = diag.util.relerr(sd["mlp_fine.rgb_linear.weight"], g["final_rgb_w"])
    print(f"trajectory {planes}: max loss deviation {dev_rel.max():.3e} (step {int(dev_rel.argmax())}), first step {dev_rel[0]:.1e}, "
          f"last {dev_rel[-1]:.1e}; final parameter norms within {dn.max():.2e}; fine rgb head weights within {dw:.2e}")
    assert dev_rel[0] < 1e-4                      # step 0 is a pure forward: the 1e-4 output bound
    assert dev_rel.max() < TRAJ_BAND[planes], (planes, dev_rel.max())
    assert dn.max() < 1e-2          # measured 1.7e-3 .. 3.3e-3 (worst: the 1e-6-scale RBK heads, which Adam moves by lr per step)


def test_reference_checkpoint_resumes_on_the_gpu(diag, tmp_path):
    """SURVEY 8f row 2 on the GPU: a file in the reference's checkpoint format (run_lushnerf.py:687-694: 108
    'module.'-prefixed keys, torch.optim.Adam state over its two parameter groups) written WITHOUT this package's
    writer, loaded into a fresh model + trainer: the train_kernel_sharp fixture is reproduced and the next Adam step
    is the one torch.optim.Adam takes from the same state."""
    import numpy as np
    from lush_nerf_amd import checkpoint as CK, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    g = diag.util.golden("train_kernel_sharp")
    lay = diag.util.golden("checkpoint_layout")
    n, Ns, Ni, naive, sharp, seed, allk = (int(x) for x in g["meta"])
    w = synth.all_weights(30, seed, sharp=True, rbk_scale=2.0e4)

    def canon(k):
        k = k[len("module."):]
        if k.startswith("blur_kernel_net.RBK."):
            return "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        if k.startswith("blur_kernel_net.view_embed_layer.") or k.startswith("dbk_view_embedding."):
            return "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        return k
    keys = [str(k) for k in lay["keys"]]
    nsd = {k: torch.from_numpy(w[canon(k)].copy()) for k in keys}
    assert len(nsd) == 108
    # optimizer state as torch writes it: parameters in the reference's order (base group, then the noise MLP)
    probe = _model(seed=1)
    noise = list(probe.mlp_noise_coarse.parameters())
    ids = set(map(id, noise))
    base = [p for p in probe.parameters() if id(p) not in ids]
    cpu_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in base + noise]
    opt = torch.optim.Adam([{"params": cpu_params[:len(base)]}, {"params": cpu_params[len(base):], "lr": 5e-4}], lr=5e-4)
    for i, p in enumerate(cpu_params):
        p.grad = torch.from_numpy(synth.normal(tuple(p.shape), 300, i)) * 1e-3
    for _ in range(3):
        opt.step()
    for gr in opt.param_groups:
        gr["lr"] = 5e-4 * 0.1 ** (2 / 250000)
    path = str(tmp_path / "000002.tar")
    torch.save({"global_step": 2, "network_state_dict": nsd, "optimizer_state_dict": opt.state_dict()}, path)

    net = _model(seed=9)                              # different weights: everything must come from the file
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, kernel_start_iter=0, allkernel_start_iter=1 << 30)
    assert CK.load_checkpoint(path, net, tr) == 2
    assert tr.global_step == 2 and tr.steps[0] == 3 and tr.steps[1] == 3
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, seed, 30).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, seed).items()}
    K = tr.K
    net.train()
    out = net(synth.H_DEF, synth.W_DEF, K, chunk=1 << 20, rays=b["rays"], rays_info={"images_idx": b["images_idx"]},
              retraw=True, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws=d, **tr.kw)
    assert diag.util.relerr(out[0], g["rgb_blur"]) < 1e-4 and diag.util.relerr(out[1], g["rgb0_blur"]) < 1e-4
    assert diag.util.relerr(out[5], g["rgb"]) < 1e-4
    before = tr.flat.param.clone()
    m0, v0 = tr.m.clone(), tr.v.clone()
    tr.step(b, 5, draws=d)
    grad = tr.flat.grad.clone()
    # the same step on the CPU with torch.optim.Adam from the file's state and the GPU's gradient
    name_of = {id(p): k for k, p in net.named_parameters()}
    gpu_params = [p for p in net.parameters() if id(p) not in set(map(id, net.mlp_noise_coarse.parameters()))] + \
        list(net.mlp_noise_coarse.parameters())
    opt2 = torch.optim.Adam([{"params": cpu_params[:len(base)]}, {"params": cpu_params[len(base):], "lr": 5e-4}], lr=5e-4)
    opt2.load_state_dict(torch.load(path, weights_only=False)["optimizer_state_dict"])
    with torch.no_grad():
        for cp, gp in zip(cpu_params, gpu_params):
            off = (gp.data_ptr() - tr.flat.param.data_ptr()) // 4
            cp.copy_(before[off:off + gp.numel()].view_as(gp).cpu())
            cp.grad = grad[off:off + gp.numel()].view_as(gp).cpu().clone()
    opt2.step()
    for cp, gp in zip(cpu_params, gpu_params):
        k = name_of[id(gp)]
        if "mlp_noise_coarse.alpha_linear" in k:
            off = (gp.data_ptr() - tr.flat.param.data_ptr()) // 4
            assert torch.equal(gp.detach().reshape(-1), before[off:off + gp.numel()]), k     # never stepped
            continue
        assert diag.util.relerr(gp.detach(), cp.detach()) < 1e-6, k


def test_trainer_over_rccl_world_size_one(diag):
    """SURVEY 8e on the box we have: Trainer(distributed=True) with backend nccl (= RCCL), world size 1 -- the
    broadcast at construction, the all-reduce of the flat gradient and grad_scale = 1/world run on RCCL -- and gives
    the step of the non-distributed trainer."""
    import torch.distributed as dist
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    import socket
    dev = torch.device("cuda:0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(64, 9).items()}
        d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(64 * 5, 64, 64, 9).items()}
        res = []
        for distributed in (True, False):
            net = _model(seed=3)
            tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64, distributed=distributed)
            assert tr.distributed == distributed and tr.world == 1
            loss = tr.step(b, 0, draws=d)
            res.append((float(loss), tr.flat.grad.clone(), tr.flat.param.clone()))
            if distributed:
                assert tr.replica_checksum() == 0.0
        assert abs(res[0][0] - res[1][0]) < 1e-6
        assert diag.util.relerr(res[0][1], res[1][1]) < 2e-4           # fp32 atomics: summation order differs run to run
        assert diag.util.relerr(res[0][2], res[1][2]) < 1e-5
    finally:
        dist.destroy_process_group()


_TWO_RANK = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
ndev = torch.cuda.device_count()
idx = rank % ndev; torch.cuda.set_device(idx); dev = torch.device("cuda", idx)
if ndev >= 2:
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
else:      # one-GPU box: both ranks share the card (RCCL refuses two ranks on one device), the collective runs over gloo
    dist.init_process_group("gloo", rank=rank, world_size=2)
from lush_nerf_amd import lib, ops, synth
from lush_nerf_amd.trainer import Trainer
import bench
lib.load()
N, Ns, Ni = 64, 64, 64
net = bench.make_model(bench.model_args(Ni), dev, ops.Precision(2, 2), seed=rank)
tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, distributed=True)
assert tr.replica_checksum() == 0.0                      # rank 1 started from another seed: the broadcast fixed it
start = tr.flat.param.clone()
batches = [{k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(N, 100 + r).items()} for r in range(2)]
draws = [{k: torch.from_numpy(v).to(dev) for k, v in synth.draws(N * 5, Ns, Ni, 100 + r).items()} for r in range(2)]
tr.step(batches[rank], 0, draws=draws[rank])
assert tr.replica_checksum() == 0.0
# the reference's DataParallel semantics (run_lushnerf.py:348, 652-661): 2 ranks x N rays == 1 rank x the concatenated 2N rays
if rank == 0:
    net1 = bench.make_model(bench.model_args(Ni), dev, ops.Precision(2, 2), seed=0)
    tr1 = Trainer(net1, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, distributed=False)
    assert torch.equal(tr1.flat.param, start)
    cat = {k: torch.cat([batches[0][k], batches[1][k]], 0) for k in batches[0]}
    dcat = {k: torch.cat([draws[0][k], draws[1][k]], 0) for k in draws[0]}
    tr1.step(cat, 0, draws=dcat)
    scale = float(tr1.flat.grad.abs().max())
    eg = float((tr.flat.grad / 2 - tr1.flat.grad).abs().max()) / scale
    ep = float((tr.flat.param - tr1.flat.param).abs().max()) / float(tr1.flat.param.abs().max())
    print(f"2 ranks vs 1 rank on the concatenated batch: gradient {eg:.1e}, parameters {ep:.1e}", flush=True)
    assert eg < 2e-4 and ep < 1e-5, (eg, ep)
    del tr1, net1
# Ranks that CHOOSE DIFFERENTLY (the live-point policy decides per rank, from its own live share: trainer.py _dense_backward_now):
# rank 0 runs the live-point backward, rank 1 the backward over all the points, in the headline mode.  Both forms add into the
# same flat gradient and every step holds exactly one collective, so: no hang, replicas identical after every step, and the pair
# still equals one rank on the concatenated batch.
H16 = ops.Precision(ops.PLANES_F16, ops.PLANES_F16)
net2 = bench.make_model(bench.model_args(Ni), dev, H16, seed=rank)
tr2 = Trainer(net2, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, distributed=True)
tr2.live_policy = "live" if rank == 0 else "dense"
start2 = tr2.flat.param.clone()
tr2.step(batches[rank], 0, draws=draws[rank])
assert tr2.replica_checksum() == 0.0
assert (tr2.live_counts()[1] > 0) == (rank == 0), (rank, tr2.live_counts())      # rank 0's march listed live points, rank 1's never did
if rank == 0:
    net3 = bench.make_model(bench.model_args(Ni), dev, H16, seed=0)
    tr3 = Trainer(net3, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, distributed=False)
    assert torch.equal(tr3.flat.param, start2)
    tr3.step(cat, 0, draws=dcat)
    scale = float(tr3.flat.grad.abs().max())
    eg = float((tr2.flat.grad / 2 - tr3.flat.grad).abs().max()) / scale
    ep = float((tr2.flat.param - tr3.flat.param).abs().max()) / float(tr3.flat.param.abs().max())
    print(f"rank 0 live-point / rank 1 dense backward vs 1 rank on the concatenated batch: gradient {eg:.1e}, parameters {ep:.1e}", flush=True)
    assert eg < 2e-4 and ep < 1e-5, (eg, ep)
# ... and under step_graph (two eager steps, then each rank captures ITS choice as two graphs around the un-captured collective)
pix = [{k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(N, 200 + 2 * s + rank).items()} for s in range(6)]
for s in range(6):
    tr2.step_graph(pix[s], 1 + s)
    assert tr2.replica_checksum() == 0.0, s
assert tr2._graph is not None and list(tr2._graph["sub"]) == [rank == 1], (rank, tr2._graph and list(tr2._graph["sub"]))
assert bool(torch.isfinite(tr2.flat.param).all()) and tr2.faults() == 0
dist.barrier()
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"r{rank}.ok"), "w").write("ok")
'''


def test_trainer_two_ranks(diag, tmp_path):
    """Two ranks, one process each: over RCCL with two GPUs; on a one-GPU box both ranks run their HIP kernels on the one card
    and the flat gradient's all-reduce goes over gloo -- the same Trainer code (broadcast at construction, one all-reduce per
    step, 1 / world folded into Adam), the reference's DataParallel semantics checked against one rank on the concatenated batch."""
    import os, subprocess, sys, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two.py"
    script.write_text(_TWO_RANK)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", port, str(script), root, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_two_ranks_rehearsal(diag, tmp_path, ranks):
    """bench.py's multi-rank control flow with real kernels on the box we have: `--gpus 2` (and 4: the driver's scaling run goes 1, 2,
    4, 8) starts the ranks itself; with
    `--rehearse-on-one-gpu` both use the one card and the collectives go over gloo.  What it pins: no rank-dependent loop count
    in front of a collective (every step of every pass holds one: warm-up, timed region, kernel-group pass, sustained cycles,
    the dense-backward comparison, the other modes), ONE JSON line from rank 0 with n_gpus = 2 and the whole-job value, the
    `n1_only` and `rehearsal` notes.  Not a measurement."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--rehearse-on-one-gpu", "--steps", "3", "--warmup", "2",
                        "--n-rand", "1024", "--no-cpu-baseline", "--also=2,2" if ranks == 2 else "--also=", "--sustained", "0.3"], capture_output=True, text=True,
                       timeout=420, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["value"] > 0 and "rehearsal" in d and "n1_only" in d
    assert d["config"]["parallelism"] == f"dp{ranks}" and d["live_points"]["share"] > 0.2
    assert d["dense_backward"]["ms_per_step"] > 0 and d["sustained"]["steps"] % 3 == 0 and (ranks != 2 or "2,2" in d["modes"])
    assert d["value_dense"] == d["dense_backward"]["value"] and d["value_at"]["live_share"] == d["live_points"]["share"]
    print(f"bench.py --gpus {ranks} rehearsed on one GPU: {d['ms_per_step']} ms/step of {ranks} time-shared ranks, all-reduce {d['allreduce_ms']} ms over gloo")


@pytest.mark.parametrize("kernel_on", [True, False])
def test_step_graph_matches_the_eager_step(diag, kernel_on):
    """Trainer.step_graph -- the whole step captured in a HIP graph, with the learning rate, Adam's step counts and the Philox
    draw counter in the device step state (lush_step_state_*) -- against Trainer.step from the same initial state over 8
    steps (two plain steps, the capture, five replays): the same losses step by step (the same draws and rates; the fp32
    atomics of the weight gradients in another order), the same counters on the host mirror, and parameters that differ
    only where Adam's m / sqrt(v) turns an atomics-order difference of a near-zero gradient into a full-size step."""
    import argparse
    import numpy as np
    from lush_nerf_amd import model as M, ops, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")

    def make():
        args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                                  N_importance=32, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                                  rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
        net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                        precision=ops.Precision(*ops.parse_planes("h,h")))
        M.load_reference_weights(net, synth.all_weights(30, 3, sharp=True))
        return Trainer(net.to(dev), synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 32, 32, kernel_start_iter=0 if kernel_on else 1 << 30,
                       allkernel_start_iter=0)

    n, steps = (64 if kernel_on else 256), 8
    bs = []
    for s in range(steps):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 5, 30, step=s).items()}
        b["target"] = torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + s))
        bs.append(b)
    torch.manual_seed(1)
    A = make()
    p0 = A.flat.param.double().clone()
    la = [float(A.step(b, s)) for s, b in enumerate(bs)]
    torch.manual_seed(1)
    B = make()
    lb = [float(B.step_graph(b, s)) for s, b in enumerate(bs)]
    assert B._graph is not None                                   # the capture happened
    assert A.steps == B.steps and A.global_step == B.global_step and A.model.hooks.draw_offset == B.model.hooks.draw_offset
    worst = max(abs(x - y) / abs(x) for x, y in zip(la, lb))
    assert worst < 2e-5, (worst, la, lb)
    pa, pb = A.flat.param.double(), B.flat.param.double()
    moved = float((pa - p0).abs().max())
    d = (pa - pb).abs()
    outliers = int((d > 2e-2 * moved).sum())
    ua, ub = pa - p0, pb - p0
    cos = float((ua @ ub) / (ua.norm() * ub.norm()))
    # (measured: 978 of 1.3 M parameters apart by more than 2 % of the largest move, the largest by 13 % of it -- what two eager
    # runs differ by as well: Adam's first steps have size lr whatever the gradient's size, so the sign of a gradient that is
    # zero up to the atomics' summation order decides a whole step)
    assert outliers < 5e-3 * pa.numel(), (outliers, float(d.max()), moved)
    assert cos > 0.999, cos
    print(f"step_graph (kernel {'on' if kernel_on else 'off'}): losses within {worst:.1e} of the eager steps; {outliers} of {pa.numel()} parameters "
          f"apart by more than 2 % of the largest 8-step move ({moved:.1e}); largest difference {float(d.max()):.1e}; cosine of the two "
          f"8-step updates {cos:.6f}")


@pytest.mark.parametrize("mode", ["interleaved", "split", "reload"])
def test_step_graph_resynchronises_with_the_host(diag, mode, tmp_path):
    """The captured step reads rate / Adam step counts / Philox counter from the device step state; whatever rewrites the
    host's counters between two replays must reach that state before the next replay:
      interleaved: step_graph x4, an eager step(), step_graph x3  == eight eager steps (same draws, same rates);
      split:       the two-graph form that ranks > 1 use ([zero, fwd, bwd] | all-reduce, not captured | [Adam, advance]) at
                   world size 1 == eight eager steps;
      reload:      a checkpoint load into a trainer that already holds a captured step: the replays continue from the
                   file's global_step, Adam counts and stored rate exactly as a fresh trainer's eager steps do.
    (Adam's bias corrections and the rate depend on the counters, the draws on the Philox offset: a stale state shows up
    in the loss from the first replay on.)"""
    import argparse
    from lush_nerf_amd import checkpoint, model as M, ops, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")

    def make(seed=3):
        args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                                  N_importance=32, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                                  rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
        net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                        precision=ops.Precision(*ops.parse_planes("h,h")))
        M.load_reference_weights(net, synth.all_weights(30, seed, sharp=True))
        return Trainer(net.to(dev), synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 32, 32, kernel_start_iter=0, allkernel_start_iter=0,
                       lrate_decay=1)          # (a fast decay, 0.1 per 1000 steps: a wrong global_step is visible in the rate)

    n, steps = 64, 8
    bs = []
    for s in range(steps):
        b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 5, 30, step=s).items()}
        b["target"] = torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + s))
        bs.append(b)
    if mode in ("interleaved", "split"):
        A = make()
        la = [float(A.step(b, s)) for s, b in enumerate(bs)]
        B = make()
        lb = []
        for s, b in enumerate(bs):
            if mode == "interleaved" and s == 5:
                assert B._graph is not None
                lb.append(float(B.step(b, s)))
            else:
                lb.append(float(B.step_graph(b, s, split=(mode == "split"))))
        assert B._graph is not None and all((g["g2"] is not None) == (mode == "split") for g in B._graph["sub"].values())
        assert A.steps == B.steps and A.global_step == B.global_step and A.model.hooks.draw_offset == B.model.hooks.draw_offset
        worst = max(abs(x - y) / abs(x) for x, y in zip(la, lb))
        assert worst < 2e-5, (mode, worst, la, lb)
        print(f"step_graph {mode}: losses within {worst:.1e} of eight eager steps")
        return
    # reload: S writes a checkpoint at global_step 700 with its own Adam state
    S = make(seed=4)
    for s in range(3):
        S.step(bs[s], s)
    S.global_step = 700
    path = str(tmp_path / "ck.tar")
    checkpoint.save_checkpoint(path, S.model, S.global_step, S)
    B = make()
    for s in range(4):
        B.step_graph(bs[s], s)
    assert B._graph is not None
    B.model.hooks.draw_offset = 0        # (the draw counter is not part of a checkpoint: both sides restart it)
    A2 = make()
    checkpoint.load_checkpoint(path, A2.model, A2)
    checkpoint.load_checkpoint(path, B.model, B)
    assert B._graph is None
    la = [float(A2.step(b, 700 + s)) for s, b in enumerate(bs[:6])]
    lb = [float(B.step_graph(b, 700 + s)) for s, b in enumerate(bs[:6])]
    assert B._graph is not None
    assert A2.steps == B.steps and A2.global_step == B.global_step == 706
    worst = max(abs(x - y) / abs(x) for x, y in zip(la, lb))
    assert worst < 2e-5, (worst, la, lb)
    dp = float((A2.flat.param.double() - B.flat.param.double()).abs().max())
    print(f"step_graph after a checkpoint load: losses within {worst:.1e} of the eager steps, parameters within {dp:.1e}")


def test_march_backward_uses_the_packed_weights_of_its_forward(diag):
    """Advisor (round 4): a forward that took the trainer's packed fragments (ops.Hooks.packed) skips the workspace copy, so a
    backward running AFTER hooks.packed was cleared (outside Trainer.step's try block, a retained graph) must still be handed
    the forward's buffers -- they are kept by the autograd node now -- instead of chaining over a never-written copy."""
    from lush_nerf_amd import ops, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    n, Ns, Ni, seed = 24, 64, 64, 9
    batch = diag.nondc_batch(n, seed).to(dev)
    draws = {k: v.to(dev) for k, v in diag.util.tdraws(n, Ns, Ni, seed).items()}
    G = diag.gpu(synth.normal((n, 3), 92))

    def grads(packed, clear):
        net = _model(Ni, diag.ops.parse_planes("h,h"), seed)
        net.train()
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni)
        if packed:
            tr._pack_weights(zero_grad=True)
            assert net.hooks.packed
        ret = net.render_rays_nonoise(batch, N_samples=Ns, retraw=True, perturb=1., N_importance=Ni, raw_noise_std=1., draws=draws)
        if clear:
            net.hooks.packed = None
        ((ret["rgb_map"] * G).sum() + (ret["rgb0"] * G).sum()).backward()
        net.hooks.packed = None
        return torch.cat([p.grad.reshape(-1) for p in list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())]).clone()
    ref = grads(False, False)
    for packed, clear in ((True, False), (True, True)):
        got = grads(packed, clear)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (packed, clear, err)


def test_second_backward_over_a_retained_graph(diag):
    """Advisor (round 4): RbkWarp / RbkWarpNdc accumulate d(r, v, w) in the zero tail of the SAVED activation rows;
    lush_rbk_mlp_bwd now leaves that tail zero again (ABI 9), so a second backward over the same graph adds exactly the first
    one's gradient once more.  Also: NoiseMlp.backward takes a [:, :3] view's base as its 4-column d_raw only when BlurMix marked
    it (column 3 = 0); any other such view gets the zero-padded copy."""
    from lush_nerf_amd import ops, synth
    dev = torch.device("cuda:0")
    g = diag.util.golden("rbk")
    n, seed = (int(x) for x in g["meta"])
    p = diag.util.params(seed, rbk_scale=3.0e5)
    b = diag.batch_of(n, seed)
    for fused in (False, True):
        tens = [diag.gpu(t).requires_grad_(True) for t in diag.rbk_tensors(p)]
        rg = diag.gpu(b["rays"]).requires_grad_(True)
        if fused:
            out, ccw, _ = ops.RbkWarpNdc.apply(rg, diag.gpu(b["images_idx"]), 4, 0.1, None, None, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF,
                                               True, 0., 1., *tens)
        else:
            out, ccw = ops.RbkWarp.apply(rg, diag.gpu(b["images_idx"]), 4, 0.1, None, None, *tens)
        loss = (out * diag.gpu(synth.normal(tuple(out.shape), 61))).sum() + (ccw * diag.gpu(synth.normal(tuple(ccw.shape), 62))).sum()
        loss.backward(retain_graph=True)
        first = [t.grad.clone() for t in tens]
        loss.backward()
        for t, f in zip(tens, first):
            assert float((t.grad - 2 * f).abs().max()) <= 2e-5 * max(float(f.abs().max()), 1e-30), fused

    class FourColumnView(torch.autograd.Function):      # an upstream op whose gradient is a [:, :3] view of a [R,4] buffer, column 3 != 0
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, gr):
            buf = torch.cat([gr, torch.full_like(gr[:, :1], 7.0)], 1).contiguous()
            return buf[:, :3]
    net = _model(64, (2, 2), 3)
    batch = diag.nondc_batch(32, 3).to(dev)
    G = diag.gpu(synth.normal((32, 3), 93))
    res = []
    for wrap in (False, True):
        net.zero_grad(set_to_none=True)
        noise = net._noise(batch, 64, False)
        ((FourColumnView.apply(noise) if wrap else noise) * G).sum().backward()
        res.append(torch.cat([p.grad.reshape(-1) for p in net.mlp_noise_coarse.parameters() if p.grad is not None]).clone())
    assert float((res[0] - res[1]).abs().max()) <= 2e-5 * float(res[0].abs().max())


def test_loss_with_one_tensor_in_both_roles(diag):
    """lush_loss_fwd_bwd with gb == NULL (a == b: no fine pass, the reference's rgb0 = rgb): the one gradient is the sum of the two
    the two-buffer form gives, the loss the same; a != b without a second buffer is refused."""
    from lush_nerf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    a = torch.rand(1000, 3, generator=g).to(dev)
    tg = torch.rand(1000, 3, generator=g).to(dev)
    l2, ga, gb = ops.train_loss_grads(a, a.clone(), tg, 0.75)
    l1, gs, none = ops.train_loss_grads(a, a, tg, 0.75)
    assert none is None and abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2)) and torch.equal(gs, ga + gb)      # (the loss is a sum of atomics: order)
    with pytest.raises(RuntimeError, match="one gradient buffer"):
        diag.lib.call("lush_loss_fwd_bwd", diag.lib.ptr(a), diag.lib.ptr(tg), diag.lib.ptr(tg), 1000, 1.0, diag.lib.ptr(torch.zeros(1, device=dev)),
                      diag.lib.ptr(gs), None, None, ops._stream())


def test_adam_multi_equals_per_segment(diag):
    """lush_adam_multi / lush_adam_state_multi (ABI 8: the active segments of the flat parameter buffer in ONE launch) give, bit
    for bit, what one lush_adam / lush_adam_state launch per segment gives -- own step count per segment, skipped segments untouched."""
    import ctypes as C
    from lush_nerf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    ends = [70001, 70001 + 3333, 70001 + 3333 + 129]
    n = ends[2]
    p0, gr, m0 = (torch.randn(n, generator=g).to(dev) for _ in range(3))
    v0 = torch.rand(n, generator=g).to(dev)
    steps = [3, 7, 1]
    for mask in (1, 3, 5, 7, 2):
        a = [t.clone() for t in (p0, m0, v0)]
        b = [t.clone() for t in (p0, m0, v0)]
        ops.adam_step_multi(a[0], gr, a[1], a[2], ends, mask, 5e-4, steps, grad_scale=0.5)
        lo = 0
        for s_, hi in enumerate(ends):
            if (mask >> s_) & 1:
                ops.adam_step(b[0][lo:hi], gr[lo:hi], b[1][lo:hi], b[2][lo:hi], 5e-4, steps[s_], grad_scale=0.5)
            lo = hi
        assert all(torch.equal(x, y) for x, y in zip(a, b)), mask
        # the device-state form: rate and bias corrections from lush_step_state (steps taken so far = steps - 1)
        state = torch.zeros(diag.lib.load().lush_step_state_bytes(), dtype=torch.uint8, device=dev)
        st = (C.c_int * 3)(*[x - 1 for x in steps])
        diag.lib.call("lush_step_state_init", diag.lib.ptr(state), C.c_ulonglong(0), 10, st, 5e-4, 250000.0, 0.9, 0.999, ops._stream())
        a = [t.clone() for t in (p0, m0, v0)]
        b = [t.clone() for t in (p0, m0, v0)]
        ops.adam_step_state_multi(a[0], gr, a[1], a[2], ends, state, mask, grad_scale=0.5)
        lo = 0
        for s_, hi in enumerate(ends):
            if (mask >> s_) & 1:
                ops.adam_step_state(b[0][lo:hi], gr[lo:hi], b[1][lo:hi], b[2][lo:hi], state, s_, grad_scale=0.5)
            lo = hi
        assert all(torch.equal(x, y) for x, y in zip(a, b)), ("state", mask)
        assert not torch.equal(a[0], p0)


@pytest.mark.parametrize("planes", ["h,h", "2,h", "2,2"])
def test_pack_plan_equals_per_network_packing(diag, planes):
    """ops.PackPlan (lush_pack_plan_*: every network of a step re-packed by ONE launch) writes, byte for byte, what
    lush_mlp_pack_for writes network by network -- the fragments of both directions and the fp32 block, for the NeRF and the
    noise net in every plane code the mode uses -- and a Trainer.step that takes its weights from the plan gives the loss of a
    step that packs inside every march."""
    from lush_nerf_amd import ops, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    prec = ops.parse_planes(planes)
    net = _model(precision=prec, seed=2)
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64)
    ent = tr._pack_entries()
    plan = ops.PackPlan(ent)
    for t in plan.buffers.values():
        t.zero_()
    # (ABI 8) the same launch clears a buffer -- the trainer's flat gradient: 16-byte aligned with a ragged end, and not aligned
    junk = torch.ones(100003, device=dev)
    plan.run(junk[:100002])
    assert float(junk[:100002].abs().max()) == 0.0 and float(junk[100002]) == 1.0
    junk.fill_(1.0)
    plan.run(junk[1:99999])
    assert float(junk[1:99999].abs().max()) == 0.0 and float(junk[0]) == 1.0 and float(junk[99999:].min()) == 1.0
    for n_, p_, tensors, variant in ent:
        nbytes = diag.lib.load().lush_mlp_packed_bytes(n_, p_)
        ref = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        st = diag.lib.mlp_struct(tensors, 8 if n_ == 0 else 4)
        import ctypes as C
        diag.lib.call("lush_mlp_pack_for", n_, p_, C.byref(st), diag.lib.ptr(ref), int(variant), ops._stream())
        got = plan.buffers[(tensors[0].data_ptr(), int(p_))]
        assert got.numel() == nbytes and torch.equal(got, ref), (n_, p_, variant)
    # a step through the plan == a step that packs per march
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(64, 9).items()}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(64 * 5, 64, 64, 9).items()}
    l1 = float(tr.step(b, 0, draws=d))
    g1 = tr.flat.grad.clone()
    net2 = _model(precision=prec, seed=2)
    tr2 = Trainer(net2, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, 64, 64)
    tr2._pack_weights = lambda zero_grad=False: tr2.flat.grad.zero_() if zero_grad else None      # every march packs for itself
    l2 = float(tr2.step(b, 0, draws=d))
    assert abs(l1 - l2) <= 1e-6 * abs(l2), (l1, l2)
    assert diag.util.relerr(g1, tr2.flat.grad) < 2e-4


def test_wide_backward_rows_match_the_half_row_kernel(diag):
    """mlp_wide_bwd_kernel against mlp_chain_bwd_half_kernel (LUSH_VARIANT_BWD_HALF), element by element on what the chain
    leaves behind: every dZ_l row the weight-gradient GEMMs read, the dZv rows with the head gradients in their extra columns,
    and d(point) -- two tiles and a ragged third (640 points: the last 128-point block of the last 256-point tile is padding).
    The two kernels round the same sums at the same places except that the 64-points-per-wave kernel pre-loads the alpha
    head's share into the accumulators (instead of adding it last) and takes the encoding derivative's sin / cos from the
    hardware: a few fp16 values land on the neighbouring grid point (measured 1.2e-4 .. 7.5e-4 of the largest entry of a layer,
    i.e. one ulp of a mid-sized value; dZv identical).  This is the test that caught, during bring-up, a v_pk_mul_f32 result lost
    in lanes 48..63 and stash rows overwritten before their store had read them."""
    import ctypes as C
    import numpy as np
    from lush_nerf_amd import lib, ops, synth
    from oracle import lush_oracle as O
    dev = torch.device("cuda:0")
    R, S = 10, 64
    w = synth.all_weights(30, 0)
    names = [f"mlp_fine.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
            [f"mlp_fine.{n}.{s}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
    tens = [torch.from_numpy(w[n]).to(dev) for n in names]
    b = synth.ray_batch(R, 1)
    batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, torch.from_numpy(b["rays"])).to(dev)
    g = torch.Generator().manual_seed(3)
    z = (torch.sort(torch.rand(R, S, generator=g), -1)[0] * 4 + 2).to(dev)
    draw = (torch.randn(R * S, 4, generator=g) * 1e-3).to(dev)
    H = ops.PLANES_F16
    pk = ops.mlp_pack(0, H, tens)
    raw, stash = ops.mlp_forward(0, H, tens, pk, batch, z, True, ops.stash_code(H, H), 0)
    P = R * S
    Ppad = (P + 255) // 256 * 256
    nbytes = lib.load().lush_mlp_dstash_bytes(0, H, P)
    st = lib.mlp_struct(tens, 8)

    def run(variant):
        ds = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        dpts = torch.zeros(P, 8, device=dev)
        lib.call("lush_mlp_bwd_chain", 0, ops.stash_code(H, H), H, lib.ptr(batch), lib.ptr(z), R, S, lib.ptr(pk), C.byref(st),
                 lib.ptr(draw), lib.ptr(stash), lib.ptr(ds), lib.ptr(dpts), variant, ops._stream())
        torch.cuda.synchronize()
        return ds.cpu().numpy(), dpts.cpu().numpy()

    a, da = run(lib.VARIANT_BWD_HALF)
    bq, db = run(0)
    al = lambda x: (x + 255) // 256 * 256
    off = 256 + al(((128 + 8) * 257 + 8 * 129) * 4)          # (lush_abi.hip dstash_layout: scale words, feature-factor scratch)
    assert np.array_equal(a[:8], bq[:8])                      # the same loss scale
    worst = 0.0
    for l in range(8):
        A = a[off:off + Ppad * 512].view(np.float16).reshape(Ppad, 256).astype(np.float32)[:P]
        B = bq[off:off + Ppad * 512].view(np.float16).reshape(Ppad, 256).astype(np.float32)[:P]
        off += al(Ppad * 512)
        assert np.isfinite(B).all() and np.abs(A).max() > 0
        err = float(np.abs(A - B).max() / np.abs(A).max())
        worst = max(worst, err)
        assert err < 2e-3, (l, err)
    A = a[off:off + Ppad * 272].view(np.float16).reshape(Ppad, 136).astype(np.float32)[:P]
    B = bq[off:off + Ppad * 272].view(np.float16).reshape(Ppad, 136).astype(np.float32)[:P]
    assert np.array_equal(A, B)                               # dZv and the head gradients: the same arithmetic
    e_pts = float(np.abs(da - db).max() / np.abs(da).max())
    assert e_pts < 2e-3, e_pts
    print(f"wide backward against the half-row kernel: dZ rows within {worst:.1e} of a layer's largest entry, dZv identical, d(point) {e_pts:.1e}")


@pytest.mark.parametrize("planes,variant", [("h,h", "HEAD_KERNEL"), ("h,h", "BWD_HALF"), ("h,h", "BWD_512"), ("h,h", "FWD_512"),
                                            ("h,h", "FWD_HALF"), ("h,h", "PE_ROWS"), ("2,1", "HEAD_KERNEL")])
def test_variants_agree(tmp_path, planes, variant):
    """The kernel variants of the C ABI (include/lush_march.h LUSH_VARIANT_*: an older kernel for the same work) against
    the product's choice: same outputs and gradients up to the rounding of the mode (the forwards keep the MFMA order
    per output; d_rgb / d_alpha travel as 16 + 16 bits with the heads folded).  Each variant runs in a child process."""
    import os, subprocess, sys
    import numpy as np
    from lush_nerf_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for on in (False, True):
        env = dict(os.environ, LUSH_PLANES=planes, LUSH_VARIANT=str(getattr(lib, "VARIANT_" + variant) if on else 0))
        out = str(tmp_path / f"ab_{int(on)}.npz")
        subprocess.run([sys.executable, os.path.join(root, "tests", "ab_worker.py"), out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    a, b = outs
    assert set(a.files) == set(b.files)
    worst = {}
    for k in a.files:
        scale = float(np.abs(b[k]).max())
        err = float(np.abs(a[k] - b[k]).max()) / max(scale, 1e-30)
        worst["raw" if k == "raw" else "grads"] = max(worst.get("raw" if k == "raw" else "grads", 0.0), err)
        # (PE_ROWS: the weight gradients read the stashed encoded rows instead of re-encoding 32 bytes per point: the same fp16
        # values bit for bit, so only the atomics' order differs.)
        # measured: backward variants leave the outputs identical; gradients 9e-7 (heads folded, fp16 hi + lo), 1.3e-5 (heads
        # folded, bf16 hi + lo), 3.4e-4 (the 256-register and the 64-points-per-wave backward chains keep d(gamma) in 16 bits; the latter also pre-loads
        # the alpha head's share into the accumulators instead of adding it last, and takes sin / cos of the encoding's derivative
        # from the hardware: 4e-7 absolute).  The FORWARD variants
        # differ in their positional encoding (mlp_wide_fwd_kernel: hardware sin / cos behind an exact range reduction,
        # 4e-7 absolute; the others: Cody-Waite + fdlibm, 7e-8): the same fp16 operand grid, but a few encodings round to
        # the neighbouring fp16 value -- raw outputs within 1.4e-4 of each other, both within 6e-4 of the fp32 oracle
        # (t_mlp_fwd gates 4e-3), gradients accordingly.
        # d(point) is per point: where a ReLU decision differs between two fp16 forwards that point's gradient moves as a
        # whole (3.3e-2 of the largest entry seen); parameter gradients average over the worker's 9 600 points (1.05e-2
        # seen; gated 3e-2, the gate the un-masked fixture comparisons use for the same reason).
        fwd_variant = variant.startswith("FWD")
        tol = (1e-3 if fwd_variant else 2e-6) if k == "raw" else ((1e-1 if k == "dpts" else 3e-2) if fwd_variant else 2e-3)
        assert np.isfinite(a[k]).all() and err < tol, (variant, k, err)
    print(f"variant {variant} ({planes}): outputs differ by {worst.get('raw', 0.0):.1e}, gradients by {worst.get('grads', 0.0):.1e}")


def test_weight_gradient_split_agrees_with_the_walk(tmp_path):
    """Round 5 experiment kept as a variant: LUSH_VARIANT_DW_SPLIT runs the grouped weight-gradient launch of a large pass as ONE
    job per workgroup on slices sized by the job's cost per point (DwGroup::per_job == 2) instead of every workgroup walking every
    job of its slice.  Same sums, other order of the fp32 atomics: 294 912 points (2 304 rays x 128 samples), every gradient."""
    import os, subprocess, sys
    import numpy as np
    from lush_nerf_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for var in (0, lib.VARIANT_DW_SPLIT):
        env = dict(os.environ, LUSH_PLANES="h,h", LUSH_VARIANT=str(var), LUSH_AB_R="2304", LUSH_AB_S="128")
        out = str(tmp_path / f"dw_{var}.npz")
        subprocess.run([sys.executable, os.path.join(root, "tests", "ab_worker.py"), out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    a, b = outs
    worst = 0.0
    for k in a.files:
        if k == "dpts" or k == "raw":
            assert np.array_equal(a[k], b[k]), k           # the forward and the chain are the same launches
            continue
        err = float(np.abs(a[k] - b[k]).max()) / max(float(np.abs(b[k]).max()), 1e-30)
        worst = max(worst, err)
        assert np.isfinite(a[k]).all() and err < 2e-5, (k, err)
    print(f"one job per workgroup against the walk: parameter gradients within {worst:.1e}")


def _march_grads(diag, variant, n, alpha_bias=None, noise=1., zero_upstream=False, seed=17, sharp=False, mode="h,h", ns=64, ni=64):
    """One kernel-on training forward + backward of the model in precision mode `mode` and the given kernel variant: (outputs,
    gradients, live counts)."""
    from lush_nerf_amd import ops, synth
    dev = torch.device("cuda:0")
    net = diag._nerf_all(ni, seed, sharp=sharp, precision=ops.Precision(*ops.parse_planes(mode), variant), rbk_scale=2.0e4)
    if alpha_bias is not None:
        with torch.no_grad():
            net.mlp_coarse.alpha_linear.bias.fill_(alpha_bias)
            net.mlp_fine.alpha_linear.bias.fill_(alpha_bias)
    b = diag.batch_of(n, seed)
    draws = {k: v.to(dev) for k, v in diag.util.tdraws(n * 5, ns, ni, seed).items()}
    rays = diag.gpu(b["rays"]).requires_grad_(True)
    net.hooks.live_acc = torch.zeros(4, dtype=torch.int64, device=dev)
    out = net(diag.H, diag.W, [[diag.F, 0, diag.W / 2], [0, diag.F, diag.H / 2], [0, 0, 1]], chunk=1 << 20, rays=rays,
              rays_info={"images_idx": diag.gpu(b["images_idx"])}, retraw=True, force_naive=False, allkernel=True,
              kernel_pixel=diag.gpu(b["fq_mask"]), perturb=1., N_importance=ni, N_samples=ns, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=noise, inference=False, near=0., far=1., draws=draws)
    loss = ops.TrainLoss.apply(out[0], out[1], diag.gpu(b["target"]))
    (loss * 0.0 if zero_upstream else loss).backward()
    assert net.read_faults() == 0
    grads = {k: (None if v.grad is None else v.grad.clone()) for k, v in net.named_parameters()}
    grads["d rays"] = rays.grad.clone()
    return [o.detach().clone() for o in (out[0], out[1], out[3], out[5])], grads, [int(x) for x in net.hooks.live_acc.tolist()]


@pytest.mark.parametrize("case", ["default-init", "default-init-large", "all-live", "all-dead", "sharp", "ragged", "ragged-samples"])
def test_live_point_march_equals_the_dense_march(diag, case):
    """Round 5: the headline mode's march keeps no stash in its forward; its backward lists the points whose d_raw row is non-zero
    (a sample whose density pre-activation the ReLU of raw2outputs clamps has alpha = 0, weight = 0, d alpha / d raw = 0:
    models/lushnerf.py:313-327), re-runs the forward with the stash on that list and chains / forms the weight gradients there.
    Against LUSH_VARIANT_DENSE_BWD (every point stashed and chained, rounds 1-4): outputs bit for bit, every gradient to the order of
    the fp32 sums.  Cases: the bench's regime (default init, raw_noise_std 1: about half the points live) at two sizes, every point live
    (density bias +5, no noise), no point live (zero upstream gradient: empty list), a sharp net."""
    from lush_nerf_amd import lib
    # (ragged: 35 marched rays -- the lists end inside a 256-point block; ragged-samples: 48 + 40 samples, rows of 88 and 48 points)
    kw = {"default-init": dict(n=96), "default-init-large": dict(n=1536), "ragged": dict(n=7), "ragged-samples": dict(n=21, ns=48, ni=40), "all-live": dict(n=48, alpha_bias=5.0, noise=0.), "all-dead": dict(n=48, zero_upstream=True),
          "sharp": dict(n=48, sharp=True)}[case]
    out_l, g_l, cnt = _march_grads(diag, 0, **kw)
    out_d, g_d, cnt_d = _march_grads(diag, lib.VARIANT_DENSE_BWD, **kw)
    assert cnt_d == [0, 0, 0, 0]                               # the dense march lists nothing
    assert all(torch.equal(a, b) for a, b in zip(out_l, out_d))
    share = (cnt[0] + cnt[2]) / max(cnt[1] + cnt[3], 1)
    assert cnt[1] == kw["n"] * 5 * (kw.get("ns", 64) + kw.get("ni", 64)) and cnt[3] == kw["n"] * 5 * kw.get("ns", 64), cnt
    if case.startswith("default-init"):
        # (96 rays: both passes list fewer than 2^18 points -- the weight-gradient launch takes ONE job per workgroup, chosen in the
        #  kernel; 1 536 rays: the fine pass lists more -- every workgroup walks the jobs of its slice)
        assert 0.3 < share < 0.7, share
        assert (cnt[0] > 262144) == (case == "default-init-large"), cnt
    elif case == "all-live":
        assert share > 0.999, share
    elif case == "all-dead":
        assert share == 0.0, share
    num = den = 0.0
    worst, wk = 0.0, ""
    for k, a in g_l.items():
        b = g_d[k]
        assert (a is None) == (b is None), k
        if a is None:
            continue
        assert torch.isfinite(a).all(), k
        if case == "all-dead":
            assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0, k
            continue
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if e > worst:
            worst, wk = e, k
        num += float((a.double() - b.double()).pow(2).sum())
        den += float(b.double().pow(2).sum())
    if case != "all-dead":
        l2 = (num / den) ** 0.5
        print(f"live-point march [{case}]: {share:.3f} of the points live; gradients against the dense march: L2 {l2:.1e}, worst tensor {wk} {worst:.1e}")
        assert l2 < 1e-5 and worst < 2e-3, (l2, wk, worst)


@pytest.mark.parametrize("mode,n", [("2,2", 96), ("2,2", 1536), ("2,1", 96), ("2,h", 96), ("h,1", 96), ("1,1", 96)])
def test_live_point_march_in_the_other_modes(diag, mode, n):
    """The live-point backward is not the headline mode's alone: every one- and two-plane mode of the 8x256 net runs it (the 128-point
    chain kernels take the same list and device-side count; the strict mode (2,2) is the fp32-equivalent figure printed next to
    `value`).  Same assertion as above against LUSH_VARIANT_DENSE_BWD, default-initialised net; 1 536 rays: the weight gradients walk."""
    from lush_nerf_amd import lib
    out_l, g_l, cnt = _march_grads(diag, 0, n, mode=mode)
    out_d, g_d, cnt_d = _march_grads(diag, lib.VARIANT_DENSE_BWD, n, mode=mode)
    assert cnt_d == [0, 0, 0, 0] and cnt[1] == n * 5 * 128 and cnt[3] == n * 5 * 64, (cnt, cnt_d)
    share = (cnt[0] + cnt[2]) / (cnt[1] + cnt[3])
    assert 0.3 < share < 0.7, share
    assert all(torch.equal(a, b) for a, b in zip(out_l, out_d))
    num = den = 0.0
    worst, wk = 0.0, ""
    for k, a in g_l.items():
        b = g_d[k]
        assert (a is None) == (b is None), k
        if a is None:
            continue
        assert torch.isfinite(a).all(), k
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if e > worst:
            worst, wk = e, k
        num += float((a.double() - b.double()).pow(2).sum())
        den += float(b.double().pow(2).sum())
    l2 = (num / den) ** 0.5
    print(f"live-point march ({mode}), {n} rays: {share:.3f} live; against the dense march: L2 {l2:.1e}, worst tensor {wk} {worst:.1e}")
    assert l2 < 1e-5 and worst < 2e-3, (l2, wk, worst)


def test_trainer_falls_back_to_the_dense_backward_when_most_points_are_live(diag):
    """The live-point march costs one forward over all the points more than the dense one: while more than Trainer.LIVE_MAX_SHARE
    of the points are live the trainer runs the dense backward (and looks again every LIVE_PROBE_EVERY steps); the share arrives
    through pinned snapshots, nothing blocks."""
    from lush_nerf_amd import ops, synth
    from lush_nerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    net = diag._nerf_all(64, 5, sharp=False, precision=ops.Precision(ops.PLANES_F16, ops.PLANES_F16, 0), rbk_scale=2.0e4)
    with torch.no_grad():
        net.mlp_coarse.alpha_linear.bias.fill_(6.0)
        net.mlp_fine.alpha_linear.bias.fill_(6.0)
    tr = Trainer(net, diag.H, diag.W, diag.F, 64, 64, kernel_start_iter=0, allkernel_start_iter=0, lrate=0.0)
    tr.LIVE_PROBE_EVERY = 4
    b = {k: diag.gpu(v) for k, v in diag.batch_of(512, 5).items()}
    modes = []
    for i in range(12):
        before = tr.live_counts()
        tr.step(b, i)
        torch.cuda.synchronize()
        after = tr.live_counts()
        modes.append("live" if after[1] > before[1] else "dense")
    assert tr.live_share is not None and tr.live_share > 0.95, tr.live_share
    assert modes[0] == "live" and "dense" in modes[1:4], modes           # switched as soon as the first share arrived
    assert modes.count("live") >= 3 and modes.count("dense") >= 6, modes  # ... and probes every fourth step
    tr.live_policy = "live"
    before = tr.live_counts()
    tr.step(b, 12)
    assert tr.live_counts()[1] > before[1]

    # the same under step_graph: one captured step per backward choice, the choice made on the host outside the capture
    def make():
        n2 = diag._nerf_all(64, 5, sharp=False, precision=ops.Precision(ops.PLANES_F16, ops.PLANES_F16, 0), rbk_scale=2.0e4)
        with torch.no_grad():
            n2.mlp_coarse.alpha_linear.bias.fill_(6.0)
            n2.mlp_fine.alpha_linear.bias.fill_(6.0)
        t = Trainer(n2, diag.H, diag.W, diag.F, 64, 64, kernel_start_iter=0, allkernel_start_iter=0)
        t.LIVE_PROBE_EVERY = 4
        return t
    A, B = make(), make()
    la, lb, mb = [], [], []
    for i in range(16):
        la.append(float(A.step(b, i)))
        torch.cuda.synchronize()
        before = B.live_counts()
        lb.append(float(B.step_graph(b, i)))
        torch.cuda.synchronize()
        mb.append("live" if B.live_counts()[1] > before[1] else "dense")
    assert B._graph is not None and set(B._graph["sub"]) == {False, True}, (B._graph and list(B._graph["sub"]), mb)
    assert mb.count("live") >= 3 and mb.count("dense") >= 8, mb
    assert A.steps == B.steps and A.global_step == B.global_step and A.model.hooks.draw_offset == B.model.hooks.draw_offset
    worst = max(abs(x - y) / abs(x) for x, y in zip(la, lb))
    print(f"step_graph under the live / dense policy ({' '.join(m[0] for m in mb)}): losses within {worst:.1e} of sixteen eager steps")
    assert worst < 5e-5, (worst, la, lb)


def test_march_through_the_c_abi_alone(diag):
    """the march is ONE C-ABI call per direction.  This test drives include/lush_march.h with ctypes and torch
    memory only -- lush_pack_rays_fwd, lush_march_workspace_bytes, lush_march_fwd, lush_march_view, lush_march_bwd -- without
    lush_nerf_amd.ops / model, on the reference fixture rays_6464_train_sharp (models/lushnerf.py:481-583 through
    render_infer :679-763): outputs at the fixture's bounds, and the backward's gradients equal to the autograd op's
    (which makes the same call) on the same inputs."""
    import ctypes as C
    import numpy as np
    from lush_nerf_amd import lib, synth
    L = lib.load()
    dev = torch.device("cuda:0")
    g = diag.util.golden("rays_6464_train_sharp")
    n, Ns, Ni, train, sharp, seed = (int(x) for x in g["meta"])
    w = synth.all_weights(diag.util.NUM_IMG, seed, sharp=bool(sharp))
    names = lambda net: [f"{net}.pts_linears.{l}.{s}" for l in range(8) for s in ("weight", "bias")] + \
        [f"{net}.{m}.{s}" for m in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear") for s in ("weight", "bias")]
    coarse = [torch.from_numpy(w[k].copy()).to(dev) for k in names("mlp_coarse")]
    fine = [torch.from_numpy(w[k].copy()).to(dev) for k in names("mlp_fine")]
    b = synth.ray_batch(n, seed, diag.util.NUM_IMG)
    rays = torch.from_numpy(b["rays"]).to(dev).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    H, W, F = synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF
    cx = float(torch.tensor(-1. / (W / (2. * F)), dtype=torch.float32))
    cy = float(torch.tensor(-1. / (H / (2. * F)), dtype=torch.float32))
    batch = torch.empty(n, 11, device=dev)
    lib.call("lush_pack_rays_fwd", lib.ptr(rays), n, 1, cx, cy, 0.0, 1.0, lib.ptr(batch), stream)
    d = {k: torch.from_numpy(v).to(dev).contiguous() for k, v in synth.draws(n, Ns, Ni, seed).items()}
    for planes, tol in (((2, 2), 1e-4), ((lib.PLANES_F16, lib.PLANES_F16), 1e-4)):
        cfg = lib.MarchCfgC(n, Ns, Ni, 1.0, 1.0, 0, 0, -1.0, planes[0], planes[1], 0, 0)
        nbytes = L.lush_march_workspace_bytes(C.byref(cfg))
        assert nbytes > 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        f = lambda *s: torch.empty(*s, device=dev)
        o = dict(rgb=f(n, 3), depth=f(n), acc=f(n), density=f(n, Ns + Ni - 1), rgb0=f(n, 3), depth0=f(n), acc0=f(n),
                 density0=f(n, Ns - 1), z_std=f(n))
        out = lib.MarchOut(*(o[k].data_ptr() for k in ("rgb", "depth", "acc", "density", "rgb0", "depth0", "acc0", "density0", "z_std")))
        dr = lib.MarchDraws(*(d[k].data_ptr() for k in ("t_rand", "noise_c", "u", "noise_f")))
        pc, pfn = lib.mlp_struct(coarse, 8), lib.mlp_struct(fine, 8)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        lib.call("lush_march_fwd", C.byref(cfg), lib.ptr(batch), C.byref(pc), C.byref(pfn), C.byref(dr), C.byref(out), lib.ptr(ws),
                 lib.ptr(flags), stream)
        assert int(flags.item()) == 0
        assert diag.util.relerr(o["rgb"], g["rgb_map"]) < tol and diag.util.relerr(o["rgb0"], g["rgb0"]) < tol
        assert diag.util.relerr(o["acc"], g["acc_map"]) < 1e-4 and diag.util.relerr(o["depth"], g["depth_map"]) < 1e-3
        off, nb = C.c_size_t(), C.c_size_t()
        lib.call("lush_march_view", C.byref(cfg), lib.VIEW_Z, C.byref(off), C.byref(nb))
        z = ws[off.value:off.value + nb.value].view(torch.float32).view(n, Ns + Ni)
        assert bool((z[:, 1:] >= z[:, :-1]).all())
        # backward: d (sum rgb + sum rgb0) / d parameters, against the autograd op on the same inputs
        gc = [torch.zeros_like(t) for t in coarse]
        gf = [torch.zeros_like(t) for t in fine]
        ones = torch.ones(n, 3, device=dev)
        go = lib.MarchGout(ones.data_ptr(), None, None, ones.data_ptr(), None, None)
        drays = torch.full((n, 11), float("nan"), device=dev)      # (ABI 7: written whole by the first pass, no zero-fill by the caller)
        sgc, sgf = lib.mlp_struct(gc, 8), lib.mlp_struct(gf, 8)
        lib.call("lush_march_bwd", C.byref(cfg), lib.ptr(batch), C.byref(pc), C.byref(pfn), C.byref(dr), C.byref(go), lib.ptr(ws),
                 C.byref(sgc), C.byref(sgf), lib.ptr(drays), stream)
        from lush_nerf_amd import ops
        cp = [t.clone().requires_grad_(True) for t in coarse]
        fp = [t.clone().requires_grad_(True) for t in fine]
        bq = batch.clone().requires_grad_(True)
        mc = ops.MarchCfg(Ns, Ni, 1., 1., precision=ops.Precision(*planes))
        res = ops.March.apply(bq, mc, d, len(cp), *cp, *fp)
        (res[0].sum() + res[7].sum()).backward()
        worst = max(diag.util.relerr(a, t.grad) for a, t in zip(gc + gf + [drays], cp + fp + [bq]))
        print(f"C ABI alone, planes {planes}: rgb_map {diag.util.relerr(o['rgb'], g['rgb_map']):.1e}; gradients vs the autograd op {worst:.1e}")
        assert worst < 2e-4, worst        # the same kernels: only the order of the fp32 atomics differs


# Bands of the long-trajectory test.  Over 300 steps the run is chaotic (the reference is fp32 torch on a CPU; here other
# summation orders, fp32 atomics in another order every run, ReLU kinks, Adam's m / sqrt(v)), so every quantity is a
# distribution.  Measured in round 3 (tools/traj_dist.py: 24 runs per kernel variant of (h,h), 12 of (2,2) and (2,h)):
#   cosine of the fine rgb head's 300-step update with the reference's: (h,h) mean 0.972 .. 0.976, sd 0.009 .. 0.016, min
#     0.937 -- the SAME for the round-2 and the round-3 kernels (the four forward x backward combinations differ by less than
#     their standard errors); (2,2) 0.982 (sd 0.008, min 0.957); (2,h) 0.983 (min 0.974)
#   largest deviation of the 25-step window means from the reference's curve: mean 0.16 .. 0.18 in every mode, max 0.32
#   fall of the loss relative to the reference's 12-fold fall: 0.94 .. 1.32 in every mode
# The modes cannot be told apart by the curve, which is the point: they train alike.  Gated on REPS runs: the mean cosine
# > 0.94 and every run > 0.88 (five standard deviations below the mean), the mean window deviation < 0.35 and every run
# < 0.5, every run's fall within a factor 1.5.  A wrong gradient does not land in these bands (a dropped layer gradient or a
# sign error leaves the cosine below 0.5 and the loss where it started); the precise per-tensor statement is the masked
# float64 check with its cosine gate (gpu_diag.masked_grad_check), which every mode passes on every fixture.
LONG_TRAJ_REPS = {"2,2": 4, "2,h": 2, "h,h": 4}      # (the fall-back mode: two runs; the suite has a 900-s limit on the driver's box)


@pytest.mark.parametrize("planes", ["2,2", "2,h", "h,h"])
def test_long_training_trajectory_follows_the_reference(diag, planes):
    """300 optimisation steps of the REAL reference (make_golden.case_trajectory_long) on teacher targets -- what a second
    weight set renders for the same rays, so the loss genuinely falls -- against Trainer.step in the fp32-equivalent mode,
    the fall-back (2,h) and the bench headline (h,h), LONG_TRAJ_REPS[mode] runs each: the windowed loss curve inside one band for
    all modes, the loss must have fallen as the reference's did, and the final fine rgb head must point the way the
    reference's does."""
    import argparse
    import numpy as np
    from lush_nerf_amd import model as M, ops, synth
    from lush_nerf_amd.trainer import Trainer
    g = diag.util.golden("train_trajectory_long")
    n, Ns, Ni, seed, steps = (int(x) for x in g["meta"])
    dev = torch.device("cuda:0")
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=Ni, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma", render_rmnearplane=80)
    ref = np.asarray(g["losses"], dtype=np.float64)
    wr = np.asarray(g["final_rgb_w"], dtype=np.float64)
    targets = torch.from_numpy(g["targets"]).to(dev)
    win = 25
    mr = ref[:steps // win * win].reshape(-1, win).mean(1)
    fall_ref = ref[-win:].mean() / ref[:win].mean()
    cosines, devs = [], []
    for rep in range(LONG_TRAJ_REPS[planes]):
        net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4),
                        precision=ops.Precision(*ops.parse_planes(planes)))
        w0 = synth.all_weights(30, seed, sharp=True, rbk_scale=2.0e4)
        M.load_reference_weights(net, w0)
        net = net.to(dev)
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, kernel_start_iter=0, allkernel_start_iter=0)
        losses = []
        for s in range(steps):
            b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, seed, 30, step=s).items()}
            b["target"] = targets[s]
            d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(n * 5, Ns, Ni, seed, step=s).items()}
            losses.append(tr.step(b, s, draws=d))
        losses = np.asarray([float(x) for x in losses])
        assert tr.faults() == 0
        mg = losses[:steps // win * win].reshape(-1, win).mean(1)
        dev_w = np.abs(mg - mr) / mr
        step_dev = np.abs(losses - ref) / ref
        wf = dict(net.state_dict())["mlp_fine.rgb_linear.weight"].detach().cpu().double().numpy()
        w_init = np.asarray(w0["mlp_fine.rgb_linear.weight"], dtype=np.float64)
        du, dr = (wf - w_init).ravel(), (wr - w_init).ravel()
        cos = float(du @ dr / (np.linalg.norm(du) * np.linalg.norm(dr)))
        fall = losses[-win:].mean() / losses[:win].mean()
        print(f"long trajectory {planes} run {rep}: reference loss {ref[:win].mean():.4f} -> {ref[-win:].mean():.4f}, here {losses[:win].mean():.4f} -> "
              f"{losses[-win:].mean():.4f}; windowed deviation max {dev_w.max():.2e} (window {int(dev_w.argmax())}), per-step max {step_dev.max():.2e}, "
              f"first step {step_dev[0]:.1e}; cosine of the fine rgb head's 300-step update with the reference's {cos:.4f}")
        assert step_dev[0] < 1e-4                      # step 0 is a pure forward: the 1e-4 output bound
        assert dev_w.max() < 0.5, (planes, dev_w)
        assert 1 / 1.5 < fall / fall_ref < 1.5, (fall, fall_ref)            # it trains as the reference trains (0.085: a 12-fold fall)
        assert cos > 0.88, cos
        cosines.append(cos)
        devs.append(float(dev_w.max()))
    assert np.mean(cosines) > 0.94, cosines
    assert np.mean(devs) < 0.35, devs
