#!/usr/bin/env python3
"""Training rays/s of the LuSh-NeRF ray-march hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one optimisation step of BASELINE config 2 per GPU: poster_lushnerf,
N_rand = 4096 input rays, N_samples 64 + N_importance 64, blur kernel (RBK/"DSK") on
(=> 20 480 marched rays, 3.93 M MLP evaluations), forward + backward + Adam, synthetic
LLFF-shaped rays resident in HBM before the timed region.  Ranks
shard rays (weak scaling); the only collective is one RCCL all-reduce of the flat
gradient buffer.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MACS_PER_EVAL = 593_408            # NeRF D=8 W=256 MLP, verified layer shapes
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# stash bytes per MLP evaluation and bf16 plane ([point][feature] rows)
BYTES_X_STASH = 2 * (128 + 8 * 256 + 256 + 128)    # gamma row, h_0..h_7, feature, views hidden
BYTES_DZ_STASH = 2 * (8 * 256 + 256 + 128)         # dZ_0..dZ_7, d feature, dZ views


def make_model(args_ns, device, precision, seed=0, num_img=30):
    from lush_nerf_amd import model as M, synth
    rbk = M.RBK(num_img, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4)
    net = M.NeRFAll(args_ns, rbk, precision=precision)
    M.load_reference_weights(net, synth.all_weights(num_img, seed))
    return net.to(device)


def model_args(n_importance):
    return argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=n_importance, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)


def cpu_baseline(n_rand, n_samples, n_importance, steps=2):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores:
    forward + loss + backward of the same kernel-on step on a bounded sample of rays."""
    from lush_nerf_amd import synth
    from oracle import lush_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # a 1-GPU box owns a 16-core share of the host (gpurun contract)
    torch.set_num_threads(cores)
    w = synth.all_weights(30, 0)
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in w.items()}
    times = []
    for s in range(steps + 1):
        b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(n_rand, 0, 30, step=s).items()}
        d = {k: torch.from_numpy(v) for k, v in synth.draws(n_rand * 5, n_samples, n_importance, 0, step=s).items()}
        t = time.perf_counter()
        out = O.forward_train(p, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, b["rays"], b["images_idx"], n_samples,
                              n_importance, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws=d)
        O.train_loss(out[0], out[1], b["target"]).backward()
        times.append(time.perf_counter() - t)
        for v in p.values():
            v.grad = None
    best = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": n_rand / best, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"N_rand={n_rand} (x5 marched), {n_samples}+{n_importance}, kernel on, fwd+bwd, "
                      f"median of {steps} steps after 1 warm-up, torch CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-rand", type=int, default=4096)
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=64)
    ap.add_argument("--planes", type=str, default="2,1", help="planes per MFMA operand fwd,bwd: h = one fp16 plane, 1..3 = bf16 planes")
    ap.add_argument("--also", type=str, default="h,1;2,2", help="second mode timed after the headline (rank 0 reports it under modes); empty to skip")
    ap.add_argument("--micro-batch", type=int, default=0, help="input rays per forward+backward slice (0 = whole batch); "
                    "bounds the activation stash for the larger BASELINE configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n-rand", type=int, default=64)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from lush_nerf_amd import lib, ops, synth
    from lush_nerf_amd.trainer import Trainer
    lib.load()

    n_batches = 4
    batches = []
    for s in range(n_batches):     # disjoint ray draws per rank; all resident before timing
        b = synth.ray_batch(a.n_rand, seed=1000 + rank, step=s)
        batches.append({k: torch.from_numpy(v).to(dev) for k, v in b.items()})

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    M = 5
    evals_step = a.n_rand * M * (a.n_samples + (a.n_samples + a.n_importance if a.n_importance else 0))

    def run_mode(pf, pb, steps, warmup):
        net = make_model(model_args(a.n_importance), dev, ops.Precision(pf, pb))
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, a.n_samples, a.n_importance, kernel_start_iter=0,
                     allkernel_start_iter=1 << 30, distributed=world > 1, micro_batch=a.micro_batch)
        for i in range(warmup):
            tr.step(batches[i % n_batches], i)
        ops.TIMER = ops.KernelTimer()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(batches[(warmup + i) % n_batches], warmup + i)
        sync()
        dt = time.perf_counter() - t0
        timer, ops.TIMER = ops.TIMER, None
        if world > 1:
            tdt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            dt = float(tdt.item())
        del tr, net
        torch.cuda.empty_cache()
        return dt, timer.summary()

    def kernel_table(groups, pf, pb, steps):
        """Per kernel group: average launch time (HIP events on the launch stream), algorithmic
        FLOP/s and algorithmic HBM bytes/s per launch (DESIGN.md section 5 gives the per-evaluation figures)."""
        sp = ops.nplanes(ops.stash_code(pf, pb))
        bytes_eval = {"mlp_fwd": sp * BYTES_X_STASH + 16 + 44 / 64, "mlp_bwd_chain": pb * BYTES_DZ_STASH + 288 + 16 + 32,
                      "mlp_bwd_weights": pb * (BYTES_X_STASH + BYTES_DZ_STASH)}
        per_prod = lambda c: {1: 1, 2: 3, 3: 6}[ops.nplanes(c)]
        mfma_mult = {"mlp_fwd": per_prod(pf), "mlp_bwd_chain": per_prod(pb), "mlp_bwd_weights": per_prod(pb)}
        kern = {}
        for g, d in groups.items():
            if g not in bytes_eval:
                continue
            avg_ms = d["ms"] / d["launches"]
            pts = d["points"] / d["launches"]
            tf = 2 * MACS_PER_EVAL * pts / (avg_ms * 1e-3) / 1e12
            gbs = bytes_eval[g] * pts / (avg_ms * 1e-3) / 1e9
            kern[g] = {"launches_per_step": d["launches"] / steps, "avg_ms": round(avg_ms, 4),
                       "ms_per_step": round(d["ms"] / steps, 3), "tflops_algorithmic": round(tf, 1),
                       "hbm_gbs_algorithmic": round(gbs, 1), "frac_mfma": round(tf / PEAK_BF16_TFLOPS, 4),
                       "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                       # plane products: the matrix pipe executes 1 / 3 / 6 MFMAs per algorithmic product
                       "frac_mfma_executed": round(tf * mfma_mult[g] / PEAK_BF16_TFLOPS, 4)}
        return kern

    pf, pb = ops.parse_planes(a.planes)
    dt, groups = run_mode(pf, pb, a.steps, a.warmup)
    others = []
    for m in [x for x in a.also.split(";") if x and x != a.planes]:
        qf, qb = ops.parse_planes(m)
        osteps = max(2, a.steps // 2)
        odt, ogroups = run_mode(qf, qb, osteps, 1)
        others.append((m, qf, qb, odt, ogroups, osteps))

    if rank == 0:
        rays_per_s = a.n_rand * world * a.steps / dt
        flop_step = 3 * 2 * MACS_PER_EVAL * evals_step          # fwd + dX + dW, algorithmic (counted once)
        kern = kernel_table(groups, pf, pb, a.steps)
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        roof = None
        if dom:
            k = kern[dom]
            hbm_bound = k["frac_hbm"] >= k["frac_mfma_executed"]      # the roof this kernel would hit first
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(f"{dom}:{a.planes}")
            roof = {"kernel": dom, "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": k["hbm_gbs_algorithmic"] if hbm_bound else k["tflops_algorithmic"],
                    "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": k["frac_hbm"] if hbm_bound else k["frac_mfma"], "traffic": traffic,
                    "executed_mfma_frac": k["frac_mfma_executed"],
                    "note": "dominant kernel group by time; achieved = algorithmic bytes (or 2*593408 FLOP) per MLP "
                            "evaluation x evaluations per launch / average launch time from HIP events on the launch stream; "
                            "the bound is the roof the kernel hits first counting the MFMAs it executes per product "
                            "(3 with 2 bf16 planes), frac stays algorithmic"}
        dtype = {1: "bf16", 2: "bf16 MFMA, operands split in 2 bf16 planes (~2^-17, fp32-equivalent outputs), fp32 accumulate",
                 3: "bf16 MFMA, 3 planes (~fp32), fp32 accumulate",
                 ops.PLANES_F16: "fp16 MFMA forward (one plane, outputs within 3e-5 of fp32), fp32 accumulate"}[pf]
        if pb != pf:
            dtype += f"; backward {pb}-plane bf16 MFMA"
        out = {
            "metric": "training rays/sec (fwd+bwd), N_samples=64+64", "value": round(rays_per_s, 1), "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"poster_lushnerf 1xMI355X N_rand={a.n_rand} N_samples={a.n_samples} "
                                   f"N_importance={a.n_importance} blur kernel (DSK/RBK) on, fwd+bwd+Adam",
                       "rays_per_gpu": a.n_rand, "marched_rays_per_gpu": a.n_rand * M, "mlp_evals_per_step": evals_step,
                       "planes_fwd": ("fp16x1" if pf == ops.PLANES_F16 else f"bf16x{pf}"), "planes_bwd": f"bf16x{pb}", "parallelism": f"dp{world}"},
            "step_tflops_algorithmic": round(flop_step * world * a.steps / dt / 1e12, 2),
            "kernels": kern, "roofline": roof,
        }
        notes = {"2,2": "every MFMA operand in 2 bf16 planes, forward AND backward (fp32-equivalent gradients)",
                 "2,1": "forward 2 bf16 planes (outputs within 5e-7 of fp32), backward plain bf16",
                 "h,1": "forward ONE fp16 plane (outputs within 3e-5 of fp32: inside the 1e-4 bound; end-to-end "
                        "gradients 4e-2..7e-2 from the reference fixtures because the larger forward rounding flips "
                        "more ReLU kinks), backward plain bf16 -- optional faster mode, not the headline"}
        if others:
            out["modes"] = {m: {"value": round(a.n_rand * world * osteps / odt, 1),
                                "ms_per_step": round(odt / osteps * 1e3, 3),
                                "kernels": kernel_table(ogroups, qf, qb, osteps), "note": notes.get(m, "")}
                            for m, qf, qb, odt, ogroups, osteps in others}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_n_rand, a.n_samples, a.n_importance)
            out["gpu_over_cpu"] = round(rays_per_s / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
