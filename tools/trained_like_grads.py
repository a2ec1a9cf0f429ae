#!/usr/bin/env python3
"""Developer aid (GPU box): gradient accuracy of a training step from the trained-like density field (synth.all_weights(trained_like=
True); N_rand 128, blur kernel on: 81 920 fine points) per precision mode, through tests/gpu_diag.t_train_bench_regime -- outputs
against the fp32 oracle, every gradient tensor against the float64 oracle run with the GPU's ReLU decisions, next to the fp32
oracle's own miss.  profiles/r06_trained_like_grads.md was written from its output."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tests.gpu_diag as D
for planes in (sys.argv[1] if len(sys.argv) > 1 else "h,h;2,h;2,2").split(";"):
    D.E2E_PLANES = D.ops.parse_planes(planes)
    D.RESULTS.clear(); D.PER_TENSOR.clear()
    D.t_train_bench_regime(n=128, seed=0, trained_like=True, min_tiles=512)
    print("##", planes, "failed gates:", [(n, float("%.3g" % e)) for n, e, t, ok in D.RESULTS if not ok])
    rows = [(k, e, f) for tag, k, e, f in D.PER_TENSOR if tag == "trained-like regime"]
    for name, sel in (("8x256 networks (mlp_coarse / mlp_fine)", lambda k: k.startswith("mlp_coarse") or k.startswith("mlp_fine")),
                      ("noise network", lambda k: k.startswith("mlp_noise")), ("blur-kernel network (mlp_rbk)", lambda k: k.startswith("mlp_rbk")),
                      ("d(rays)", lambda k: k == "grad_rays")):
        g = sorted((e, f, k) for k, e, f in rows if sel(k))
        if g:
            print(f"   {name}: {len(g)} tensors, median {g[len(g) // 2][0]:.1e}, worst {g[-1][0]:.1e} ({g[-1][2]}; fp32 oracle there {g[-1][1]:.1e}); "
                  f"fp32 oracle median {sorted(f for e, f, k in g)[len(g) // 2]:.1e}")
