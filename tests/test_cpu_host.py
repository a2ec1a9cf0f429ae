"""CPU-side tests: C-ABI exports, host logic, deterministic generator, 2-rank gloo data-parallel plumbing.
No compute call into the HIP library is made here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from lush_nerf_amd import lib, synth
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    lib.build()
    so = ctypes.CDLL(lib.SO_PATH)
    header = open(os.path.join(ROOT, "include", "lush_march.h")).read()
    declared = set(re.findall(r"\b(lush_[a-z0-9_]+)\s*\(", header))
    declared -= {"lush_stream_t"}
    assert len(declared) >= 30
    missing = [n for n in sorted(declared) if not hasattr(so, n)]
    assert not missing, missing
    assert set(lib.EXPORTS) <= declared | {"lush_last_error"}
    l = lib.load()
    assert l.lush_abi_version() == 1
    # pure host queries (no device work)
    p1, p3 = l.lush_mlp_packed_bytes(0, 1), l.lush_mlp_packed_bytes(0, 3)
    assert p1 > 2 * 593408 * 2 and 2.9 * p1 < p3 < 3 * p1      # fragments scale with planes, the fp32 bias block does not
    assert l.lush_mlp_packed_bytes(7, 1) == 0
    assert l.lush_mlp_stash_bytes(0, 2, 2, 64) > 0 and l.lush_mlp_stash_bytes(0, 2, 1, 65) == l.lush_mlp_stash_bytes(0, 2, 1, 128)
    assert 0 < l.lush_mlp_stash_bytes(0, 2, 0, 128) < l.lush_mlp_stash_bytes(0, 2, 1, 128) < l.lush_mlp_stash_bytes(0, 2, 2, 128)


def test_product_refuses_cpu_tensors():
    from lush_nerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.PackRays.apply(torch.zeros(4, 3, 2), 8, 8, 10.0, True, 0., 1.)


def test_product_does_not_import_the_oracle():
    for dp, dn, fn in os.walk(os.path.join(ROOT, "lush_nerf_amd")):
        for f in fn:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|import_module\(.oracle|__import__\(.oracle", src, re.M), f


def test_generator_is_deterministic_and_reference_shaped():
    a, b = synth.all_weights(30, 5), synth.all_weights(30, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert sum(v.size for v in a.values()) == 1301993          # flat gradient size
    assert a["mlp_coarse.pts_linears.5.weight"].shape == (256, 319)
    assert a["mlp_coarse.views_linears.0.weight"].shape == (128, 283)
    assert a["mlp_noise_coarse.pts_linears.0.weight"].shape == (128, 63)
    assert float(np.abs(a["mlp_rbk.r_linear.weight"]).max()) < 1.5e-6
    r1, r2 = synth.ray_batch(64, 1, step=0), synth.ray_batch(64, 1, step=1)
    assert not np.array_equal(r1["rays"], r2["rays"]) and r1["rays"].shape == (64, 3, 2)
    u = synth.uniform01(1000, 3)
    assert 0 <= u.min() and u.max() < 1


def test_model_mirror_has_reference_state_dict_keys():
    import argparse
    from lush_nerf_amd import model as M
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=64, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)
    net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4))
    sd = net.state_dict()
    assert len(sd) == 108 and sum(p.numel() for p in net.parameters()) == 1301993
    for k in ("mlp_coarse.pts_linears.0.weight", "mlp_fine.rgb_linear.bias", "mlp_noise_coarse.alpha_linear.weight",
              "blur_kernel_net.RBK.r_linear.weight", "dbk_view_embedding.view_embed_layer.weight",
              "mlp_rbk.view_embed_linears.3.bias", "blur_kernel_net.view_embed_layer.view_embed_layer.weight"):
        assert k in sd, k
    M.load_reference_weights(net, synth.all_weights(30, 2))
    assert np.array_equal(net.mlp_fine.pts_linears[5].weight.detach().numpy(),
                          synth.all_weights(30, 2)["mlp_fine.pts_linears.5.weight"])
    with pytest.raises(NotImplementedError):
        args2 = argparse.Namespace(**{**vars(args), "multires": 6})
        M.NeRFAll(args2, None)


def test_flat_params_views():
    from lush_nerf_amd.trainer import FlatParams
    lin1, lin2 = torch.nn.Linear(3, 4), torch.nn.Linear(4, 2)
    w0 = lin1.weight.detach().clone()
    fp = FlatParams([list(lin1.parameters()), list(lin2.parameters()) + [lin1.weight]])   # alias is de-duplicated
    assert fp.numel == 3 * 4 + 4 + 4 * 2 + 2 and fp.segments == [(0, 16), (16, 26)]
    assert torch.equal(lin1.weight.detach(), w0)
    fp.param[:12] += 1.0
    assert torch.equal(lin1.weight.detach(), w0 + 1.0)
    (lin2(lin1(torch.ones(5, 3))).sum()).backward()
    assert float(fp.grad.abs().sum()) > 0 and lin1.weight.grad.data_ptr() == fp.grad.data_ptr()


_DIST_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from lush_nerf_amd.trainer import FlatParams
from lush_nerf_amd import synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
fp = FlatParams([list(net.parameters())])
# every rank draws its own rays (disjoint seeds), SURVEY 8e
b = synth.ray_batch(32, seed=1000 + rank)
x = torch.from_numpy(b["rays"]).reshape(32, 6)
loss = ((net(x) - torch.from_numpy(b["target"])) ** 2).mean()
loss.backward()
local = fp.grad.clone()
dist.all_reduce(fp.grad)                       # the one collective of the step
gathered = [torch.zeros_like(local) for _ in range(world)]
dist.all_gather(gathered, local)
assert torch.allclose(fp.grad, sum(gathered)), "all-reduce != sum of per-rank gradients"
fp.param -= 0.1 * fp.grad / world              # redundant update on every rank
chk = [torch.zeros_like(fp.param) for _ in range(world)]
dist.all_gather(chk, fp.param)
assert all(torch.equal(chk[0], c) for c in chk), "replicas diverged"
assert not torch.equal(gathered[0], gathered[1]), "ranks must see different rays"
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"rank{rank}.ok"), "w").write("ok")     # stdout of the two ranks interleaves
'''


def test_data_parallel_allreduce_two_ranks_gloo(tmp_path):
    import socket
    script = tmp_path / "dp.py"
    script.write_text(_DIST_SCRIPT)
    with socket.socket() as sk:                      # a free rendezvous port (a fixed one can sit in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout[-1000:]


def _small_trainer(seed=0):
    import argparse
    from lush_nerf_amd import model as M
    from lush_nerf_amd.trainer import Trainer
    args = argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=64, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)
    net = M.NeRFAll(args, M.RBK(30, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4))
    M.load_reference_weights(net, synth.all_weights(30, seed))
    return net, Trainer(net, 640, 1120, 1000.0)


def test_checkpoint_interop_with_reference_layout(tmp_path):
    """the file we write has exactly the reference's keys/shapes, its optimizer state loads
    into a torch.optim.Adam built the way the reference builds it, and a round trip restores everything."""
    from lush_nerf_amd import checkpoint as CK
    g = util.golden("checkpoint_layout")
    net, tr = _small_trainer(3)
    tr.steps = [7, 5, 0]
    tr.global_step = 7
    tr.m.copy_(torch.from_numpy(synth.normal((tr.m.numel(),), 1)))
    tr.v.copy_(torch.from_numpy(synth.uniform((tr.v.numel(),), 0, 1, 2)))
    path = str(tmp_path / "000007.tar")
    CK.save_checkpoint(path, net, 7, tr)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"global_step", "network_state_dict", "optimizer_state_dict"}
    assert list(ck["network_state_dict"].keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in ck["network_state_dict"].values()] == [str(s) for s in g["shapes"]]
    osd = ck["optimizer_state_dict"]
    assert [len(gr["params"]) for gr in osd["param_groups"]] == [int(x) for x in g["group_sizes"]]
    # loads into an optimizer constructed exactly as run_lushnerf.py:359-371 does
    noise = list(net.mlp_noise_coarse.parameters())
    ids = set(map(id, noise))
    base = [p for p in net.parameters() if id(p) not in ids]
    assert [p.numel() for p in base] == [int(x) for x in g["group0_numel"]]
    assert [p.numel() for p in noise] == [int(x) for x in g["group1_numel"]]
    opt = torch.optim.Adam([{"params": base}, {"params": noise, "lr": 5e-4}], lr=5e-4)
    opt.load_state_dict(osd)
    st = opt.state[net.mlp_fine.pts_linears[3].weight]
    off = (net.mlp_fine.pts_linears[3].weight.data_ptr() - tr.flat.param.data_ptr()) // 4
    assert torch.equal(st["exp_avg"].reshape(-1), tr.m[off:off + 65536]) and float(st["step"]) == 7
    assert net.mlp_noise_coarse.alpha_linear.weight not in opt.state            # never stepped -> no state
    # round trip into a fresh model / trainer
    net2, tr2 = _small_trainer(4)
    assert CK.load_checkpoint(path, net2, tr2) == 7
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    assert tr2.steps == [7, 5, 0] and tr2.global_step == 7
    a, b = tr.flat.segments[1]
    assert torch.equal(tr2.m[:b], tr.m[:b]) and torch.equal(tr2.v[:b], tr.v[:b])
