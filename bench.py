#!/usr/bin/env python3
"""Training rays/s of the LuSh-NeRF ray-march hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
  or plainly: with WORLD_SIZE unset, bench.py starts the N ranks itself as a child torch.distributed.run
  BEFORE anything touches the GPU, and exits with the child's code.  It refuses to run when the RCCL world
  size differs from --gpus or when fewer than N devices are present.

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
# What the chip SUSTAINS on random fp16 operands with nothing but the MFMAs and their fragment reads in the loop (the forward's
# skeleton: v_mfma_f32_32x32x16_f16, A re-read from LDS, B and the accumulators in registers; tools/micro/mfma_shape.hip,
# profiles/r06_mfma_shape.txt): the clock is held at 1.73 GHz there (2.4 GHz on all-zero operands).  Quoted next to the datasheet
# peak the way MI355X_MICROARCH.md quotes 6.3 TB/s next to HBM's 8: `frac` stays against the datasheet figure.
SUSTAINED_F16_TFLOPS = 1546.0
# stash bytes per MLP evaluation and bf16 plane ([point][feature] rows)
# A 1- or 2-plane backward keeps neither the feature activations nor their gradients (the feature layer is linear: its
# weight gradients follow from dZv^T h_7, FeatFactorArgs in csrc/lush_mlp.h); the 3-plane reference mode stashes both.
def BYTES_X_STASH(planes, reencode=False):
    # gamma row (one fp16 plane, product kernels: 32 bytes of point + view direction that the weight gradients re-encode),
    # h_0..h_7, [feature], views hidden
    return (32 if reencode else 2 * 128) + 2 * (8 * 256 + (256 if planes >= 3 else 0) + 128)


def BYTES_DZ_STASH(planes):
    # dZ_0..dZ_7, [d feature], dZ views, [one plane: the head gradients as 8 more columns of the dZ-views rows]
    return 2 * (8 * 256 + (256 if planes >= 3 else 0) + 128 + (8 if planes == 1 else 0))


def make_model(args_ns, device, precision, seed=0, num_img=30, trained_like=False):
    from lush_nerf_amd import model as M, synth
    rbk = M.RBK(num_img, 64, 4, 64, 1, 32, 1, 32, 1, 32, 3, 3, [4], True, 0.1, 4)
    net = M.NeRFAll(args_ns, rbk, precision=precision)
    M.load_reference_weights(net, synth.all_weights(num_img, seed, trained_like=trained_like))
    return net.to(device)


def model_args(n_importance):
    return argparse.Namespace(blur_model_type="dpnerf", multires=10, multires_views=4, i_embed=0, use_viewdirs=True,
                              N_importance=n_importance, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                              rgb_activate="sigmoid", sigma_activate="relu", tone_mapping_type="gamma",
                              render_rmnearplane=80)


def _host_cpu():
    """(threads to use, CPU model, note): the cores this process may run on = min(scheduler affinity, cgroup CPU quota).
    A 1-GPU box owns a share of a large host: its affinity mask can list every core of the host while the cgroup
    quota grants 16; a torch thread pool sized by the mask is then oversubscribed ~10x (measured: 146 s per step)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    affinity = cores
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())        # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = max(1, min(cores, int(quota + 0.5)))
    if os.environ.get("LUSH_CPU_THREADS"):
        cores = int(os.environ["LUSH_CPU_THREADS"])
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, model, f"affinity {affinity} cores, cgroup quota {quota if quota is not None else 'none'}"


def cpu_baseline(n_rand, n_samples, n_importance, steps=5, warmup=2, c1_steps=5, progress=None):
    """The oracle (CPU restatement of the reference path, parity-pinned against the reference's own outputs)
    timed on this box's host cores, as SURVEY 8d / BASELINE.md section 3 plan it: all cores the process may use,
    >= 2 warm-up and >= 5 timed steps, median; (a) the poster 64+64 kernel-on training step at N_rand 512
    (forward + loss + backward) and (b) BASELINE config 1 (N_rand 256, 32+0, naive, entry render_infer)."""
    from lush_nerf_amd import synth
    from oracle import lush_oracle as O
    cores, model, cpu_note = _host_cpu()
    torch.set_num_threads(cores)
    if progress:
        progress(f"  host: {model}, using {cores} threads ({cpu_note})")
    w = synth.all_weights(30, 0)
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in w.items()}

    def timed(fn, n_warm, n_timed):
        ts = []
        for s in range(n_warm + n_timed):
            t = time.perf_counter()
            fn(s)
            ts.append(time.perf_counter() - t)
            if progress:
                progress(f"  cpu step {s}: {ts[-1]:.2f} s")
            for v in p.values():
                v.grad = None
        ts = sorted(ts[n_warm:])
        return ts[len(ts) // 2]

    def kernel_on(s):
        b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(n_rand, 0, 30, step=s).items()}
        d = {k: torch.from_numpy(v) for k, v in synth.draws(n_rand * 5, n_samples, n_importance, 0, step=s).items()}
        out = O.forward_train(p, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, b["rays"], b["images_idx"], n_samples,
                              n_importance, force_naive=False, allkernel=True, kernel_pixel=b["fq_mask"], draws=d)
        O.train_loss(out[0], out[1], b["target"]).backward()

    def config1(s):
        b = {k: torch.from_numpy(v) for k, v in synth.ray_batch(256, 0, 30, step=s).items()}
        d = {k: torch.from_numpy(v) for k, v in synth.draws(256, 32, 0, 0, step=s).items()}
        batch = O.pack_rays(synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, b["rays"])
        ret, _ = O.render_rays(p, batch, 32, retraw=True, perturb=1., N_importance=0, raw_noise_std=1., draws=d)
        rgb = O.tonemap(ret["rgb_map"])
        (0.5 * torch.mean((rgb - b["target"]) ** 2) + 0.5 * torch.mean(torch.abs(rgb - b["target"]))).backward()

    t_k = timed(kernel_on, warmup, steps)
    t_1 = timed(config1, warmup, c1_steps)
    return {"value": n_rand / t_k, "unit": "rays/s", "cores": cores, "cpu_model": model, "cpu_share": cpu_note, "kind": "port",
            "sample": f"N_rand={n_rand} (x5 marched), {n_samples}+{n_importance}, blur kernel on, forward+loss+backward, "
                      f"median of {steps} steps after {warmup} warm-up, torch CPU fp32, {cores} threads",
            "s_per_step": round(t_k, 3),
            "config1": {"value": 256 / t_1, "unit": "rays/s", "s_per_step": round(t_1, 4),
                        "sample": f"BASELINE config 1: N_rand=256, 32+0, naive, entry render_rays/render_infer, forward+loss+"
                                  f"backward, median of {c1_steps} after {warmup} warm-up"}}


def measure_traffic(argv_tail, kernel_key, progress=None, timeout=240):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this very command under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domains: MI355X_MICROARCH.md, TCC slots), each
    a fresh process started as a CHILD (the parent keeps running; nothing is exec'ed).  FETCH_SIZE is doubled (gfx950 reports
    half of a wide coalesced stream); both counters are in KiB.  Returns a dict, or None when rocprofv3 is missing or a pass
    fails (the caller then quotes profiles/pmc_traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None           # this process already runs under a profiler: no nested tool
    out = {}
    with tempfile.TemporaryDirectory(prefix="lush_pmc_", dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp", MASTER_PORT=str(29600 + os.getpid() % 300))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            cmd = [prof, "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), *argv_tail]
            if progress:
                progress(f"  rocprofv3 --pmc {c} pass of the same command (child process)")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            except (subprocess.TimeoutExpired, OSError):
                return None
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            total, launches = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if kernel_key in row["Kernel_Name"] and row["Counter_Name"] == c:
                    total += float(row["Counter_Value"]) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
                    launches.add(row["Dispatch_Id"])
            if not launches:
                return None
            out[c] = total / len(launches)
            out[c + "_launches"] = len(launches)
    return {"hbm_bytes_per_launch": round(out["FETCH_SIZE"] + out["WRITE_SIZE"]), "fetch_bytes_per_launch": round(out["FETCH_SIZE"]),
            "write_bytes_per_launch": round(out["WRITE_SIZE"]), "launches_counted": out["FETCH_SIZE_launches"],
            "source": "measured in this run: child `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same command "
                      "(FETCH_SIZE x2: gfx950 correction), average over the kernel's launches"}


def traffic_child_args(a):
    """Arguments of the child runs measure_traffic profiles: the SAME workload as this run (sizes, micro-batch, mode, variant,
    library), two steps, nothing else timed."""
    return ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--also=", "--extra=", "--no-kernel-pass", "--no-traffic", "--sustained", "0", "--no-dense",
            "--planes", a.planes, "--config", a.config, "--variant", str(a.variant), "--n-rand", str(a.n_rand), "--n-samples", str(a.n_samples),
            "--n-importance", str(a.n_importance), "--micro-batch", str(a.micro_batch)] + (["--so", a.so] if a.so else [])


CONFIGS = {   # BASELINE.json configs: input rays per GPU, N_samples, N_importance, blur kernel on, micro-batch
    "C1": dict(n_rand=256, ns=32, ni=0, kernel=False, micro=0),
    "C2": dict(n_rand=4096, ns=64, ni=64, kernel=True, micro=0),
    "C3": dict(n_rand=8192, ns=64, ni=64, kernel=True, micro=0),
    "C5": dict(n_rand=16384, ns=128, ni=128, kernel=True, micro=4096),
}


def self_launch(a):
    """`python bench.py --gpus N` with no launcher: start N fresh ranks as a CHILD torch.distributed.run.  Nothing in
    this process has touched the GPU yet (torch.cuda.device_count() does not initialise it on this image), and the
    process is never replaced: it waits for the child and exits with its code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and not getattr(a, "rehearse_on_one_gpu", False):
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_dev} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-rand", type=int, default=4096)
    ap.add_argument("--n-samples", type=int, default=64)
    ap.add_argument("--n-importance", type=int, default=64)
    ap.add_argument("--planes", type=str, default="h,h", help="16-bit planes per MFMA operand fwd,bwd: h = one fp16 plane (backward: "
                    "loss-scaled), 1..3 = bf16 planes.  Default = the fastest mode that passes every parity gate (tests/test_gpu_parity.py)")
    ap.add_argument("--also", type=str, default="2,2;2,1;2,h", help="other modes timed after the headline (rank 0 reports them under modes; "
                    "2,2 = the strict fp32-equivalent mode, printed next to value); empty to skip")
    ap.add_argument("--micro-batch", type=int, default=0, help="input rays per forward+backward slice (0 = whole batch); "
                    "bounds the activation stash for the larger BASELINE configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="developer A/B: lib.VARIANT_* bits, an older kernel for the same work "
                    "(include/lush_march.h); 0 = the product's choice, the only value a reported line may carry")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the second, kernel-group-timed pass over the steps (profiling "
                    "runs: the process then launches exactly what the timed region launches; the line carries no `kernels` / `roofline`)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic (two child rocprofv3 --pmc passes of this "
                    "command, rank 0, N = 1 only); quote profiles/pmc_traffic.json instead")
    ap.add_argument("--so", type=str, default=None, help="developer A/B: another build of the library (tools/build_variant.py); the line "
                    "then carries its path under `library` -- a reported line has none")
    ap.add_argument("--sustained", type=float, default=3.0, help="seconds of back-to-back headline steps behind the timed region, reported "
                    "under `sustained` (0 to skip)")
    ap.add_argument("--no-dense", action="store_true", help="skip the comparison run with the backward over all the points (dense_backward)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="developer: run the N ranks of --gpus N on the GPUs that exist "
                    "(rank r on device r mod count), collectives over gloo -- a rehearsal of the multi-rank control flow on a one-GPU box; "
                    "the line says so under `rehearsal` and is not a measurement")
    ap.add_argument("--cpu-n-rand", type=int, default=512, help="input rays of the CPU baseline's kernel-on step (SURVEY 8d: 512)")
    ap.add_argument("--config", type=str, default="C2", choices=["C2", "C3", "C5"],
                    help="BASELINE config timed as the headline workload (C2 = the one the metric is quoted on)")
    ap.add_argument("--extra", type=str, default=None, help="other BASELINE configs reported under extra_configs "
                    "(default at 1 GPU: C1,C3,C5,eval; none on multi-GPU runs); empty to skip")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.config != "C2":
        c = CONFIGS[a.config]
        a.n_rand, a.n_samples, a.n_importance = c["n_rand"], c["ns"], c["ni"]
        a.micro_batch = a.micro_batch or c["micro"]

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)                                   # never returns
    # stdout carries ONE line, the JSON: whatever the libraries underneath print to file descriptor 1 (RCCL's version banner, gloo's
    # connection notes) goes to stderr instead -- descriptor 1 is pointed at stderr now and the line is written to the saved one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the reported n_gpus must be the RCCL world size")
    if a.rehearse_on_one_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py needs an MI355X per rank (no CPU fallback of the product path): rank {rank} has no device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the RCCL group exists at every N (at N = 1 the all-reduce is a one-rank collective): the timed path is the same code
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if a.rehearse_on_one_gpu:      # RCCL refuses two ranks on one device: the rehearsal's collectives run over gloo
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if dist.get_world_size() != a.gpus:
        raise SystemExit(f"RCCL world size {dist.get_world_size()} != --gpus {a.gpus}")

    from lush_nerf_amd import lib, ops, synth
    from lush_nerf_amd.trainer import Trainer
    if a.so:
        lib.use_library(a.so)
    lib.load()
    t_start = time.perf_counter()

    def progress(msg):       # stderr only (stdout carries the ONE JSON line); also keeps gpurun's idle watchdog fed
        if rank == 0:
            print(f"[bench {time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    n_batches = 4
    poses = torch.from_numpy(synth.poses(30, 1000 + rank)).to(dev)

    def make_batches(n_rand):
        """(view, pixel) draws + targets, disjoint per rank, resident in HBM before timing; the rays
        themselves are generated on the device every step (lush_gen_rays)."""
        out = []
        for s in range(n_batches):
            b = synth.pixel_batch(n_rand, seed=1000 + rank, step=s)
            b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
            b["c2w"] = poses
            out.append(b)
        return out

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    M = 5

    def evals_per_step(n_rand, ns, ni, kernel=True):
        return n_rand * (M if kernel else 1) * (ns + (ns + ni if ni else 0))

    def run_mode(pf, pb, steps, warmup, cfg=None, sustained_s=0.0, variant=None, trained_like=False):
        """Time `steps` optimisation steps of one BASELINE config in one precision mode (max over ranks).  sustained_s > 0: after
        the timed region, the same steps back to back for at least that many seconds (the timed region of the driver's
        `--steps 20` is a third of a second on a chip that runs at its power cap: this is its steady-state neighbour)."""
        cfg = cfg or dict(n_rand=a.n_rand, ns=a.n_samples, ni=a.n_importance, kernel=True, micro=a.micro_batch)
        batches = make_batches(cfg["n_rand"])
        net = make_model(model_args(cfg["ni"]), dev, ops.Precision(pf, pb, a.variant if variant is None else variant), trained_like=trained_like)
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, cfg["ns"], cfg["ni"], kernel_start_iter=0,
                     allkernel_start_iter=1 << 30, distributed=True, micro_batch=cfg["micro"])
        if trained_like:
            # The density field is HELD: every kernel of the step runs (both marches, loss, backward, all-reduce, Adam), Adam with a
            # zero rate.  From fresh Adam moments the first updates are +-lrate whatever the gradient, and through the x3000 density
            # head that moves the field from 0.23 live to 0.67 within three steps (tools/trained_like_probe.py): the workload would
            # time a transient of the optimiser, not a trained scene's density distribution.
            tr.lrate = 0.0
        for i in range(warmup):
            tr.step(batches[i % n_batches], i)
        sync()
        # The state the timed region starts from.  The step's cost depends on the model since round 5 (the live-point march skips the
        # backward of the points whose gradient is exactly zero, and the model keeps training on the synthetic targets), so the two
        # passes behind the timed region -- kernel groups, sustained -- REPLAY the timed region's steps from this state: same
        # parameters, same optimiser moments, same Philox draws, the same live points.
        snap = (tr.flat.param.clone(), tr.m.clone(), tr.v.clone(), list(tr.steps), tr.global_step, net.hooks.draw_offset)

        def restore():
            tr.flat.param.copy_(snap[0]); tr.m.copy_(snap[1]); tr.v.copy_(snap[2])
            tr.steps, tr.global_step, net.hooks.draw_offset = list(snap[3]), snap[4], snap[5]
        # the share of MLP evaluation points whose backward ran (the live-point march, include/lush_march.h): the trainer counts
        # them on the device (one small add per march, no synchronisation); read around the timed region
        live0 = tr.live_counts()
        tr.allreduce_events = []
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(batches[(warmup + i) % n_batches], warmup + i)
        sync()
        dt = time.perf_counter() - t0
        ar_ms = sum(a_.elapsed_time(b_) for a_, b_ in tr.allreduce_events) / max(len(tr.allreduce_events), 1)
        tr.allreduce_events = None

        def live_share(c0, c1):
            fl, fa, cl, ca = (b_ - a_ for a_, b_ in zip(c0, c1))
            return None if fa + ca == 0 else {"share": round((fl + cl) / (fa + ca), 4), "fine": round(fl / fa, 4) if fa else None,
                                               "coarse": round(cl / ca, 4) if ca else None}
        live = live_share(live0, tr.live_counts())
        dense_steps = int(getattr(tr, "_dense_steps", 0)) if getattr(tr, "_dense_now", False) else 0      # (what the trainer's policy chose)
        # kernel groups: the SAME steps once more (state restored) with HIP events around each MLP kernel group on the launch stream.
        # The timed region above runs the march as one C-ABI call per direction (lush_march_fwd / lush_march_bwd); with the timer set
        # the same kernels are launched group by group through the piecewise entry points.
        timer = ops.KernelTimer()
        if not a.no_kernel_pass:
            restore()
            net.hooks.timer = timer
            for i in range(steps):
                tr.step(batches[(warmup + i) % n_batches], warmup + i)
            sync()
            net.hooks.timer = None
        sustained = None
        if sustained_s > 0:
            cdt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(cdt, op=dist.ReduceOp.MAX)      # (every rank runs the same number of cycles: each step holds a collective)
            cycles = max(1, int(sustained_s / max(float(cdt.item()), 1e-6)) + 1)
            live1 = tr.live_counts()
            sync()
            t1 = time.perf_counter()
            for c in range(cycles):
                restore()
                for i in range(steps):
                    tr.step(batches[(warmup + i) % n_batches], warmup + i)
            sync()
            n_sus = cycles * steps
            sdt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
            sustained = {"steps": n_sus, "seconds": round(float(sdt.item()), 3), "ms_per_step": round(float(sdt.item()) / n_sus * 1e3, 3),
                         "value": round(cfg["n_rand"] * world * n_sus / float(sdt.item()), 1), "unit": "rays/s",
                         "live_points": live_share(live1, tr.live_counts()),
                         # (the replay's check on itself: the same states and draws give the same live points, to the count)
                         "same_live_points_as_timed_region": live_share(live1, tr.live_counts()) == live,
                         "note": "the timed region's %d steps replayed back to back for >= %g s: each cycle restores the model, the optimiser "
                                 "moments and the draw counter to the state the timed region started from (three device copies of 5 MB) "
                                 "and runs the same steps -- the same live points (live_points), the same work; what differs from the "
                                 "timed region is how long the chip has been at its power cap" % (steps, sustained_s)}
        tdt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)       # the slowest rank's clock
        dt = float(tdt.item())
        fault = tr.faults()
        if fault:
            raise SystemExit(f"numerical fault during the timed steps: {net.fault_names(fault)}")
        del tr, net, batches
        torch.cuda.empty_cache()
        summ = timer.summary()
        summ["_allreduce_ms"] = ar_ms
        summ["_sustained"] = sustained
        summ["_live"] = live
        summ["_dense_steps"] = dense_steps
        return dt, summ

    def run_c1(pf, pb, steps, warmup):
        """BASELINE config 1 (N_rand 256, 32+0, naive): NeRFAll.forward cannot take N_importance = 0 (it indexes
        extras['rgb0']), so the entry is render_infer, as in the reference's own CPU-runnable case."""
        c = CONFIGS["C1"]
        from lush_nerf_amd.trainer import BlobBatch
        batches = [BlobBatch(b) for b in make_batches(c["n_rand"])]      # per-step inputs in one buffer: one copy per replay
        net = make_model(model_args(0), dev, ops.Precision(pf, pb, a.variant)).train()
        tr = Trainer(net, synth.H_DEF, synth.W_DEF, synth.FOCAL_DEF, c["ns"], 0, kernel_start_iter=1 << 30, distributed=True)

        # the step itself is Trainer.step_coarse_only (render_infer at 32 + 0, one summed loss gradient, Adam): the same function
        # tests/test_gpu_parity.py::test_unverified_gradient_paths_in_the_headline_mode[t_train_c1] checks against the oracle, eagerly and
        # as the graph replayed here
        for i in range(warmup):
            tr.step_coarse_only(batches[i % n_batches])
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step_coarse_only(batches[(warmup + i) % n_batches])
        sync()
        dt_eager = time.perf_counter() - t0
        # the same step captured in a HIP graph: one launch per step, the batch copied into the graph's tensors first
        replay, static = tr.capture_coarse_only(batches[0])

        def run(i):
            static.load(batches[i % n_batches])
            replay()
        for i in range(warmup):
            run(i)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            run(warmup + i)
        sync()
        dt = time.perf_counter() - t0
        del tr, net, replay
        return dt, dt_eager

    def run_eval(pf, steps):
        """NeRFAll.forward(poses=...) -> render_path, one 640x1120 pose per step, forward only."""
        net = make_model(model_args(64), dev, ops.Precision(pf, 1)).eval()
        K = [[synth.FOCAL_DEF, 0, synth.W_DEF / 2], [0, synth.FOCAL_DEF, synth.H_DEF / 2], [0, 0, 1]]
        rk = dict(perturb=False, N_importance=64, N_samples=64, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
                  inference=True, near=0., far=1.)
        net(synth.H_DEF, synth.W_DEF, K, chunk=1 << 15, poses=poses[:1], render_kwargs=dict(rk))
        sync()
        t0 = time.perf_counter()
        net(synth.H_DEF, synth.W_DEF, K, chunk=1 << 15, poses=poses[1:1 + steps], render_kwargs=dict(rk))
        sync()
        dt = time.perf_counter() - t0
        del net
        torch.cuda.empty_cache()
        return dt

    def kernel_table(groups, pf, pb, steps):
        """Per kernel group: average launch time (HIP events on the launch stream), algorithmic
        FLOP/s and algorithmic HBM bytes/s per launch (DESIGN.md section 5 gives the per-evaluation figures)."""
        sp = ops.nplanes(ops.stash_code(pf, pb))
        pbn = ops.nplanes(pb)
        ree = pf == ops.PLANES_F16 and pbn == 1 and not (a.variant & (lib.VARIANT_FWD_HALF | lib.VARIANT_FWD_512 | lib.VARIANT_PE_ROWS))
        # the forward over ALL the points of a live-point march keeps no stash: 16 bytes of raw output and 4 of depth per point, the
        # ray's 44 bytes once per ~96 samples, and (added per launch below) the weight stream: every XCD's L2 fetches the network's
        # 1.19 MB of fp16 fragments from HBM once per launch (8 x 1.19 MB; the CUs re-read it from L2 per tile)
        bytes_eval = {"mlp_fwd_all": 16 + 4 + 44 / 96,
                      "mlp_fwd": sp * BYTES_X_STASH(sp, ree) + 16 + 44 / 64,
                      "mlp_bwd_chain": pbn * BYTES_DZ_STASH(pbn) + 288 + 16 + 32,
                      "mlp_bwd_weights": pbn * (BYTES_X_STASH(pbn, ree) + BYTES_DZ_STASH(pbn))}
        per_prod = lambda c: {1: 1, 2: 3, 3: 6}[ops.nplanes(c)]
        mfma_mult = {"mlp_fwd_all": per_prod(pf), "mlp_fwd": per_prod(pf), "mlp_bwd_chain": per_prod(pb), "mlp_bwd_weights": per_prod(pb)}
        kern = {}
        for g, d in groups.items():
            if g not in bytes_eval or not isinstance(d, dict):
                continue
            avg_ms = d["ms"] / d["launches"]
            pts = d["points"] / d["launches"]
            tf = 2 * MACS_PER_EVAL * pts / (avg_ms * 1e-3) / 1e12
            per_launch = 8 * 2 * MACS_PER_EVAL * (1 if g in ("mlp_fwd_all", "mlp_fwd", "mlp_bwd_chain") else 0)      # the weight stream, once per XCD
            gbs = (bytes_eval[g] * pts + per_launch) / (avg_ms * 1e-3) / 1e9
            kern[g] = {"launches_per_step": d["launches"] / steps, "avg_ms": round(avg_ms, 4),
                       "ms_per_step": round(d["ms"] / steps, 3), "tflops_algorithmic": round(tf, 1),
                       "hbm_gbs_algorithmic": round(gbs, 1), "frac_mfma": round(tf / PEAK_BF16_TFLOPS, 4),
                       "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                       # plane products: the matrix pipe executes 1 / 3 / 6 MFMAs per algorithmic product
                       "frac_mfma_executed": round(tf * mfma_mult[g] / PEAK_BF16_TFLOPS, 4)}
        return kern

    pf, pb = ops.parse_planes(a.planes)
    evals_step = evals_per_step(a.n_rand, a.n_samples, a.n_importance)
    progress(f"headline mode {a.planes}: {a.warmup} warm-up + {a.steps} timed steps on {world} rank(s)")
    dt, groups = run_mode(pf, pb, a.steps, a.warmup, sustained_s=a.sustained)
    progress(f"headline: {dt / a.steps * 1e3:.2f} ms/step" + (f"; sustained {groups['_sustained']['ms_per_step']:.2f} ms/step over "
                                                               f"{groups['_sustained']['seconds']:.1f} s" if groups.get("_sustained") else ""))
    # the same configuration with the backward over ALL the points (LUSH_VARIANT_DENSE_BWD: rounds 1-4; what the step costs
    # whatever the data): reported next to the headline whenever the headline's march is the live-point one
    dense = None
    if ops.live_backward(ops.Precision(pf, pb, a.variant), None) and not a.no_dense:
        dsteps = max(4, a.steps // 2)
        ddt, _ = run_mode(pf, pb, dsteps, 2, variant=a.variant | lib.VARIANT_DENSE_BWD)
        dense = {"value": round(a.n_rand * world * dsteps / ddt, 1), "unit": "rays/s", "ms_per_step": round(ddt / dsteps * 1e3, 3), "steps": dsteps,
                 "note": "LUSH_VARIANT_DENSE_BWD: the forward stashes every point and the backward runs over all of them (rounds 1-4): "
                         "the step's cost independent of the data"}
        progress(f"dense backward (all points): {dense['ms_per_step']:.2f} ms/step")
    others = []
    for m in [x for x in a.also.split(";") if x and x != a.planes]:
        qf, qb = ops.parse_planes(m)
        osteps = max(2, a.steps // 2)
        odt, ogroups = run_mode(qf, qb, osteps, 1)
        progress(f"mode {m}: {odt / osteps * 1e3:.2f} ms/step")
        others.append((m, qf, qb, odt, ogroups, osteps))
    extra_names = [x for x in (a.extra if a.extra is not None else ("trained_like,C1,C3,C5,eval" if world == 1 else "")).split(",") if x]
    extras = {}
    for name in extra_names:       # the other BASELINE configs with the SAME kernels and precision mode as the headline
        progress(f"extra config {name}")
        if name == "eval":
            n_pose = 2
            edt = run_eval(pf, n_pose)
            extras["eval"] = {"value": round(n_pose * synth.H_DEF * synth.W_DEF * world / edt, 1), "unit": "rays/s (forward only)",
                              "s_per_pose": round(edt / n_pose, 4),
                              "workload": "NeRFAll.forward(poses) -> render_path, 640x1120 rays per pose, 64+64, chunk 32768"}
        elif name == "trained_like":
            # The headline workload from a density field shaped like a trained scene's (synth.all_weights(trained_like=True): empty
            # space strongly negative, "surfaces" strongly positive, the fine network agreeing with the coarse one) instead of the
            # initialisation's sigma ~ 0, where the unit noise alone decides which half of the samples is live
            # (configs/poster_lushnerf:14, 21: 100 000 iterations at raw_noise_std = 1e0: most of a run is NOT at initialisation).
            st = max(4, a.steps // 2)
            tdt, tgroups = run_mode(pf, pb, st, 3, trained_like=True)
            extras["trained_like"] = {"value": round(a.n_rand * world * st / tdt, 1), "unit": "rays/s", "ms_per_step": round(tdt / st * 1e3, 3), "steps": st,
                                      "live_points": tgroups.get("_live"),
                                      "backward": "dense (the policy's choice)" if tgroups.get("_dense_steps") else "live points (the policy's choice)",
                                      "kernels": kernel_table(tgroups, pf, pb, st),
                                      "workload": "BASELINE config 2 from synth.all_weights(trained_like=True): sigma = 3000 w.h + 20 (positive in ~13 % of "
                                                  "the volume, tens where it is; dead under the unit noise in ~85 %), fine network = coarse network; the field "
                                                  "is held (Adam runs with a zero rate: bench.py run_mode says why); same rays, draws, kernels and mode as `value`"}
        elif name == "C1":
            st = 20
            cdt, cdt_eager = run_c1(pf, pb, st, 3)
            extras["C1"] = {"value": round(256 * world * st / cdt, 1), "unit": "rays/s", "ms_per_step": round(cdt / st * 1e3, 3),
                            "ms_per_step_eager": round(cdt_eager / st * 1e3, 3),
                            "workload": "N_rand=256, 32+0, naive, entry render_infer, fwd+bwd+Adam, 8 192 MLP evaluations: the step captured in a "
                                        "HIP graph (rate / Adam step / draw counter in the device step state) and replayed; ms_per_step_eager = "
                                        "the same step launched kernel by kernel from Python"}
        elif name in CONFIGS and name != a.config:
            c = CONFIGS[name]
            st = 3 if name == "C3" else 2
            cdt, cgroups = run_mode(pf, pb, st, 1, c)
            extras[name] = {"value": round(c["n_rand"] * world * st / cdt, 1), "unit": "rays/s",
                            "ms_per_step": round(cdt / st * 1e3, 3), "micro_batch": c["micro"] or c["n_rand"],
                            "mlp_evals_per_step": evals_per_step(c["n_rand"], c["ns"], c["ni"]),
                            "workload": f"N_rand={c['n_rand']} per GPU, {c['ns']}+{c['ni']}, blur kernel on, fwd+bwd+Adam",
                            "kernels": kernel_table(cgroups, pf, pb, st)}

    if rank == 0:
        rays_per_s = a.n_rand * world * a.steps / dt
        flop_step = 3 * 2 * MACS_PER_EVAL * evals_step          # fwd + dX + dW, algorithmic (counted once)
        kern = kernel_table(groups, pf, pb, a.steps)
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        roof = None
        if dom:
            k = kern[dom]
            hbm_bound = k["frac_hbm"] >= k["frac_mfma_executed"]      # the roof this kernel would hit first
            traffic, traffic_source = None, None
            dom_kernel = {"mlp_bwd_weights": "dw_group_kernel<true", "mlp_fwd": "mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 1>",
                          "mlp_fwd_all": "mlp_wide_fwd_kernel<lush::NetT<256, 8, 5>, 0>", "mlp_bwd_chain": "mlp_wide_bwd_kernel"}.get(dom)
            if world == 1 and not a.no_traffic and dom_kernel and a.planes == "h,h":
                tail = traffic_child_args(a)
                torch.cuda.empty_cache()
                traffic = measure_traffic(tail, dom_kernel, progress)
                if traffic:
                    traffic_source = traffic.pop("source")
                    traffic["algorithmic_bytes_per_launch"] = round(k["hbm_gbs_algorithmic"] * 1e9 * k["avg_ms"] * 1e-3)
            if traffic is None:
                tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(tpath):     # per-launch HBM bytes from the separate rocprofv3 --pmc passes of profiles/collect.sh
                    traffic = json.load(open(tpath)).get(f"{dom}:{a.planes}")
                traffic_source = "profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes; not re-measured in this run)"
            # the same kernel against BOTH roofs and BOTH byte models: `design_bytes` = what this design moves per evaluation in
            # that kernel (it stashes dZ as well as X), `survey_stash_bytes` = SURVEY 8d's 4.35 KB per
            # evaluation (the X stash alone, read once by this pass)
            pts_launch = groups[dom]["points"] / groups[dom]["launches"]
            design_b = k["hbm_gbs_algorithmic"] * 1e9 * k["avg_ms"] * 1e-3 / pts_launch
            survey_b = 4352.0
            survey_gbs = survey_b * pts_launch / (k["avg_ms"] * 1e-3) / 1e9
            both = {"frac_mfma": k["frac_mfma"], "frac_hbm": k["frac_hbm"],
                    "design_bytes_per_eval": round(design_b), "frac_hbm_design_bytes": k["frac_hbm"]}
            if dom != "mlp_fwd_all":      # (the forward over all the points keeps no stash: SURVEY 8d's stash bytes are not its traffic)
                both.update({"survey_stash_bytes_per_eval": int(survey_b), "frac_hbm_survey_stash_bytes": round(survey_gbs / PEAK_HBM_GBS, 4)})
            if traffic and traffic.get("hbm_bytes_per_launch"):
                both["traffic_over_design"] = round(traffic["hbm_bytes_per_launch"] / (design_b * pts_launch), 3)
                if dom != "mlp_fwd_all":
                    both["traffic_over_survey"] = round(traffic["hbm_bytes_per_launch"] / (survey_b * pts_launch), 3)
            roof = {"kernel": dom, "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": k["hbm_gbs_algorithmic"] if hbm_bound else k["tflops_algorithmic"],
                    "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": k["frac_hbm"] if hbm_bound else k["frac_mfma"], "traffic": traffic,
                    "traffic_source": traffic_source,
                    "executed_mfma_frac": k["frac_mfma_executed"], **both,
                    # the datasheet MFMA peak is not reachable on random data: the clock is held down under MFMA load
                    # (profiles/r06_pair_proto.md section 2); frac_of_sustained = achieved over what a bare MFMA + LDS-fragment loop sustains
                    "sustained_peak": SUSTAINED_F16_TFLOPS if (not hbm_bound and pf == ops.PLANES_F16) else None,
                    "frac_of_sustained_peak": round(k["tflops_algorithmic"] / SUSTAINED_F16_TFLOPS, 4) if (not hbm_bound and pf == ops.PLANES_F16) else None,
                    "note": "dominant kernel group by time; achieved = algorithmic bytes (or 2*593408 FLOP) per MLP "
                            "evaluation x evaluations per launch / average launch time from HIP events on the launch stream; "
                            "the bound is the roof the kernel hits first counting the MFMAs it executes per product "
                            "(3 with 2 bf16 planes), frac stays algorithmic"}
        dtype = {1: "bf16", 2: "bf16 MFMA, operands split in 2 bf16 planes (~2^-17, fp32-equivalent outputs), fp32 accumulate",
                 3: "bf16 MFMA, 3 planes (~fp32), fp32 accumulate",
                 ops.PLANES_F16: "f16 MFMA forward (one fp16 plane per operand, fp32 accumulate; render outputs within 3e-5 of the fp32 reference)"}[pf]
        if pb == ops.PLANES_F16:
            dtype += "; backward fp16 MFMA (one plane, per-launch power-of-two loss scale), fp32 accumulate"
        elif pb != pf:
            dtype += f"; backward {pb}-plane bf16 MFMA"
        out = {
            "metric": "training rays/sec (fwd+bwd), N_samples=64+64", "value": round(rays_per_s, 1), "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"poster_lushnerf {world}xMI355X N_rand={a.n_rand}/GPU N_samples={a.n_samples} "
                                   f"N_importance={a.n_importance} blur kernel (DSK/RBK) on, fwd+bwd+Adam, "
                                   f"one RCCL all-reduce of the flat gradient (BASELINE config {a.config if world == 1 else '4' if a.config == 'C2' else a.config})",
                       "rays_per_gpu": a.n_rand, "marched_rays_per_gpu": a.n_rand * M, "mlp_evals_per_step": evals_step,
                       "planes_fwd": ("fp16x1" if pf == ops.PLANES_F16 else f"bf16x{pf}"), "planes_bwd": ("fp16x1 (loss-scaled)" if pb == ops.PLANES_F16 else f"bf16x{pb}"), "parallelism": f"dp{world}"},
            # the share of MLP evaluation points whose d_raw row is non-zero in the timed steps: the live-point march (round 5) re-runs
            # the forward with the stash and runs the gradient chain and the weight gradients on THOSE points only (a point whose
            # density pre-activation the ReLU of raw2outputs clamps -- with raw_noise_std = 1 and a near-zero density, half of them --
            # has an exactly zero gradient); null: the backward ran over all the points
            "live_points": groups.get("_live"),
            "dense_backward": dense,
            # ... the same two figures where a reader of `value` looks first: `value` is (mode, live share); `value_dense` is the step
            # whose cost does not depend on the model's state (the backward over all the points: the figure comparable with rounds 1-4)
            "value_at": {"mode": a.planes, "live_share": (groups.get("_live") or {}).get("share")},
            "value_dense": dense["value"] if dense else None,
            "ms_per_step_dense": dense["ms_per_step"] if dense else None,
            "step_tflops_algorithmic": round(flop_step * world * a.steps / dt / 1e12, 2),
            # ... and what the kernels executed: one forward over all the points + (forward, chain, weight gradients) on the live ones
            "step_tflops_executed": round((1 + 3 * (groups["_live"]["share"] if groups.get("_live") else 2 / 3)) / 3 * flop_step * world * a.steps / dt / 1e12, 2),
            # whole step against the survey's two ceilings: 3 x 1 186 816 FLOP per evaluation on the dense
            # bf16 MFMA peak, and the stash written once + read once at 4.35 KB per evaluation on the HBM peak
            "step_frac_mfma": round(flop_step * a.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            # ... the same with the executed flops (a skipped dead point earns nothing here)
            "step_frac_mfma_executed": round((1 + 3 * (groups["_live"]["share"] if groups.get("_live") else 2 / 3)) / 3 * flop_step * a.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            "step_frac_hbm_survey_stash": round(2 * 4352 * evals_step * a.steps / dt / 1e9 / PEAK_HBM_GBS, 4),
            "kernels": kern, "roofline": roof,
            # the one collective of a step (HIP events around dist.all_reduce of the 5.2 MB flat gradient on the launch
            # stream, mean over the timed steps): what an N > 1 run adds to the N = 1 step
            "allreduce_ms": round(groups.get("_allreduce_ms", 0.0), 4),
            "sustained": groups.get("_sustained"),
            "kernel_timing": "HIP events around each MLP kernel group on the launch stream, over the same steps run once more "
                             "group by group (the timed region makes one lush_march_fwd / lush_march_bwd call per march and has no "
                             "events inside; every kernel runs alone on the one stream in both passes)",
        }
        notes = {"2,h": "forward 2 bf16 planes, backward ONE loss-scaled fp16 plane (11-bit operands at the bf16 backward's cost)",
                 "h,h": "forward ONE fp16 plane (outputs within 3e-5 of fp32), backward ONE loss-scaled fp16 plane",
                 "2,2": "every MFMA operand in 2 bf16 planes, forward AND backward (fp32-equivalent gradients)",
                 "2,1": "forward 2 bf16 planes (outputs within 5e-7 of fp32), backward plain bf16",
                 "h,1": "forward ONE fp16 plane (outputs within 3e-5 of fp32: inside the 1e-4 bound; end-to-end "
                        "gradients 4e-2..7e-2 from the reference fixtures because the larger forward rounding flips "
                        "more ReLU kinks), backward plain bf16 -- optional faster mode, not the headline"}
        if a.so:
            out["library"] = os.path.abspath(a.so)
        if a.rehearse_on_one_gpu:
            out["rehearsal"] = (f"{world} ranks time-sharing {torch.cuda.device_count()} GPU(s), collectives over gloo: the multi-rank control "
                                "flow of this script, NOT a measurement")
        if others:
            out["modes"] = {m: {"value": round(a.n_rand * world * osteps / odt, 1),
                                "ms_per_step": round(odt / osteps * 1e3, 3),
                                "live_points": ogroups.get("_live"),
                                "kernels": kernel_table(ogroups, qf, qb, osteps), "note": notes.get(m, "")}
                            for m, qf, qb, odt, ogroups, osteps in others}
            if "2,2" in out["modes"]:     # the strict figure sits next to `value`, not only under modes
                out["value_strict_2_2"] = out["modes"]["2,2"]["value"]
                out["ms_per_step_strict_2_2"] = out["modes"]["2,2"]["ms_per_step"]
                out["live_points_strict_2_2"] = out["modes"]["2,2"]["live_points"]      # (the live-point backward runs in every one- / two-plane mode)
        if extras:
            out["extra_configs"] = extras
            tl = extras.get("trained_like")
            if tl and tl.get("live_points") and out["live_points"] and abs(tl["live_points"]["share"] - out["live_points"]["share"]) > 0.05:
                # the step against the live share, from the two workloads of THIS run (same box, same process): ms = a + b x share
                s1, m1, s2, m2 = out["live_points"]["share"], out["ms_per_step"], tl["live_points"]["share"], tl["ms_per_step"]
                b_ = (m1 - m2) / (s1 - s2)
                out["ms_per_step_vs_live_share"] = {"a_ms": round(m1 - b_ * s1, 3), "b_ms_per_unit_share": round(b_, 3),
                                                    "from": f"this run's two workloads: share {s1} -> {m1} ms (value), share {s2} -> {m2} ms (extra_configs.trained_like)",
                                                    "dense_ms": out["ms_per_step_dense"],
                                                    "note": "a = the forward over all the points + everything outside the MLPs; the dense backward is the cheaper "
                                                            "one above share = (dense_ms - a) / b (profiles/r05_live_points.md: 5.28 + 14.54 x share over 600 steps)"}
        if world > 1:      # a SCALE line explains what it leaves out
            out["n1_only"] = ("cpu_baseline, roofline.traffic (live rocprofv3 --pmc passes) and extra_configs are N = 1 quantities: see the "
                              "N = 1 line of the same round (BENCH_rNN.json); roofline.traffic here quotes profiles/pmc_traffic.json")
        if not a.no_cpu_baseline and world == 1:
            progress(f"CPU baseline (oracle on the host cores, N_rand {a.cpu_n_rand} kernel-on + config 1)")
            out["cpu_baseline"] = cpu_baseline(a.cpu_n_rand, 64, 64, progress=progress)
            progress("CPU baseline done")
            out["gpu_over_cpu"] = round(rays_per_s / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
