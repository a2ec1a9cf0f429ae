"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel and the end-to-end
path, through the C ABI, against the CPU oracle and the reference-made golden fixtures.

Tolerances (normalised max error = max|a-b| / max|b|):
  * forward outputs (rgb_map, rgb0, acc_map, tone-mapped colours, loss): 1e-4  -- the
    north-star bound; depth_map / z_std 1e-3 / 2e-3 because sample_pdf amplifies 1-ulp cdf
    differences by 1/pdf (a property of the reference algorithm in fp32);
  * kernel-level backward vs torch autograd with the GPU's own ReLU decisions: 2e-5 (3 planes),
    2e-4 (2 planes), 3e-2 (1 plane = plain bf16);
  * end-to-end gradients vs the reference fixture: 3e-2 (ReLU-kink conditioning).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SECTIONS = ["t_zgrid_pack", "t_gen_rays", "t_composite", "t_sample", "t_mlp_fwd", "t_mlp_ragged", "t_mlp_bwd", "t_rbk", "t_mix",
            "t_march_e2e", "t_train_e2e"]


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
    (measured 2.7e-5) at a third of the MFMA work; the price is gradient noise from ReLU kinks flipped
    by the larger forward rounding (measured 4e-2..7e-2 vs the reference fixtures, gated at 1e-1) and a
    looser z_std (sample_pdf conditioning), which is why it is not the bench headline."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes("h,1"))
    diag.RESULTS.clear()
    diag.t_march_e2e()
    res = {n: e for n, e, t, ok in diag.RESULTS}
    for n, e in res.items():
        if n.endswith("rgb_map") or n.endswith("rgb0") or n.endswith("noise_rgb") or n.endswith("acc_map"):
            assert e < 1e-4, (n, e)
        if n.endswith("depth_map"):
            assert e < 1e-3, (n, e)
        if n.endswith("z_std"):
            assert e < 3e-2, (n, e)
    diag.RESULTS.clear()
    diag.t_train_e2e()
    for n, e, t, ok in diag.RESULTS:
        if "worst grad" in n or n.endswith("grad_rays"):
            assert e < 1e-1, (n, e)
        elif "grad-None" in n:
            assert ok, n
        else:
            assert e < 1e-4, (n, e)


@pytest.mark.parametrize("planes", ["2,1"])
def test_headline_mode_end_to_end(diag, planes, monkeypatch):
    """The bench headline mode (2 planes forward, bf16 backward): forward within 1e-4 of the reference
    fixtures, end-to-end gradients inside the same 3e-2 gate as the fp32-equivalent mode."""
    monkeypatch.setattr(diag, "E2E_PLANES", diag.ops.parse_planes(planes))
    for section in ("t_march_e2e", "t_train_e2e"):
        diag.RESULTS.clear()
        getattr(diag, section)()
        bad = [(n, e, t) for n, e, t, ok in diag.RESULTS if not ok]
        assert not bad, bad[:8]


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


def test_full_size_properties(diag):
    """BASELINE config 2 size (20 480 marched rays, 64+64): size-independent properties."""
    from lush_nerf_amd import ops, synth
    from oracle import lush_oracle as O
    dev = torch.device("cuda:0")
    net = _model().train()
    R, Ns, Ni = 20480, 64, 64
    b = synth.ray_batch(R, 5)
    batch = ops.PackRays.apply(torch.from_numpy(b["rays"]).to(dev), synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF,
                               True, 0., 1.)
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.draws(R, Ns, Ni, 5).items()}
    kw = dict(N_samples=Ns, N_importance=Ni, perturb=1., raw_noise_std=1., retraw=True)
    with torch.no_grad():
        cfg = ops.MarchCfg(Ns, Ni, 1., 1., precision=ops.Precision(2, 2), want_grad=False)
        coarse, fine = net.mlp_coarse.tensors(), net.mlp_fine.tensors()
        out = ops.March.apply(batch, cfg, d, len(coarse), *coarse, *fine)
        out2 = ops.March.apply(batch, cfg, d, len(coarse), *coarse, *fine)
    rgb, depth, acc, density, raw, weights, z = out[:7]
    assert all(torch.equal(a, b_) for a, b_ in zip(out, out2)), "forward is not deterministic"
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
    assert diag.util.relerr(rgb[idx], ref["rgb_map"]) < 1e-4
    assert diag.util.relerr(out[7][idx], ref["rgb0"]) < 1e-4


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
    """BASELINE configs 3 (8192 rays, 64+64) and 5 (16 384 rays, 128+128, the HBM stress case) run one
    optimisation step with bounded memory (micro-batches of 4096 input rays) and give finite results."""
    from lush_nerf_amd import synth
    from lush_nerf_amd.trainer import Trainer
    import bench
    dev = torch.device("cuda:0")
    n, Ns, Ni = (8192, 64, 64) if cfg == "C3" else (16384, 128, 128)
    net = bench.make_model(bench.model_args(Ni), dev, diag.ops.Precision(2, 1))
    tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, Ns, Ni, micro_batch=4096)
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.ray_batch(n, 8).items()}
    torch.cuda.reset_peak_memory_stats()
    loss = tr.step(b, 0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(tr.flat.param).all())
    assert float(tr.flat.grad.abs().max()) > 0
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"{cfg}: loss {float(loss):.4f}, peak torch memory {peak:.1f} GiB")
    assert peak < 200


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
