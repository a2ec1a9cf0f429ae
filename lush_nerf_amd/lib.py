"""Loader / builder for liblush_march.so (the hand-written HIP kernels + C ABI).

The shared object is built IN-TREE by ``build()`` (hipcc --offload-arch=gfx950)
and loaded with ctypes.  There is no fallback: if the library is missing or a
call fails the product raises.  Signatures mirror include/lush_march.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.path.join(HERE, "liblush_march.so")      # (developer tools load another build with use_library(path) BEFORE load())
SOURCES = ["lush_march.hip", "lush_mlp.hip", "lush_mlp_chain.hip", "lush_mlp_wide.hip", "lush_mlp_wide_bwd.hip", "lush_abi.hip", "lush_march_abi.hip"]
HEADERS = ["lush_common.h", "lush_mlp.h", "lush_mlp_dev.h", "lush_mlp_wide.h", "lush_host.h", os.path.join("..", "..", "include", "lush_march.h")]

_lib = None


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def use_library(path: str):
    """Developer tools only (tools/*.py: ablation / profiling builds): load() will open `path` instead of the in-tree product."""
    global SO_PATH, _lib
    if _lib is not None:
        raise RuntimeError("lib.use_library: a library is already loaded")
    SO_PATH = os.path.abspath(path)


def build(force: bool = False, verbose: bool = False, extra_flags: Sequence[str] = (), out: str = None, audit: bool = True) -> str:
    """Compile the HIP sources for gfx950 into lush_nerf_amd/liblush_march.so and audit the result (isa_check).

    The product build takes nothing from the environment.  `extra_flags` / `out` are for developer builds (tools/build_variant.py:
    -DLUSH_PROF, the -DLUSH_ABL_* timing ablations) and refuse to write the product's path."""
    product = out is None
    if extra_flags and product:
        raise ValueError("lib.build: extra compiler flags need an explicit `out` path (the in-tree product is built without any)")
    target = SO_PATH if product else os.path.abspath(out)
    if product and not force and not needs_build():
        return SO_PATH
    # -ffp-contract=off: fp32 VALU expressions round op by op like the reference's torch ops
    # (o + d*z must not become one FMA: a 1-ulp point error is amplified x512 by the encoding).
    # one hipcc -c per source, in parallel (the chain kernels alone take ~1 min), then one link
    import re
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *extra_flags]
    # per file: the 64-points-per-wave kernels keep their accumulators in VGPRs (the VALU converts them; AGPR-resident
    # accumulators cost one v_accvgpr_read per value) and let the B-operand buffers, which only MFMAs read, go to AGPRs
    per_file = {"lush_mlp_wide.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "lush_mlp_wide_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
    usage = {}
    with tempfile.TemporaryDirectory(prefix="lush_build_") as tmp:
        def compile_one(f):
            obj = os.path.join(tmp, os.path.splitext(f)[0] + ".o")
            cmd = [_hipcc(), *flags, *per_file.get(f, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            # resource usage per kernel (informational; build_report() prints it).  Scalar spills are legal compiler behaviour;
            # what made round 3's spilled builds fault is checked on the emitted code below (isa_check rule R1).
            names = re.findall(r"Function Name: (\S+)", r.stderr)
            sg = [int(m) for m in re.findall(r"SGPRs Spill: (\d+)", r.stderr)]
            vg = [int(m) for m in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
            sc = [int(m) for m in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
            if f in per_file and not (names and len(names) == len(sg) == len(vg) == len(sc)):
                raise RuntimeError(f"{f}: no kernel-resource-usage remarks parsed (hipcc output format changed?)")
            for n, a_, b_, c_ in zip(names, sg, vg, sc):
                usage[n] = {"file": f, "sgpr_spills": a_, "vgpr_spills": b_, "scratch_bytes_per_lane": c_}
            return obj
        with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", target + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    if audit:
        # The hazards hipcc does not pad for `asm volatile` statements, checked on what was actually emitted (DESIGN.md
        # section 4, "Hazards"): a build in which register allocation or scheduling happened to create one never reaches a GPU.
        from . import isa_check
        n, found = isa_check.check_shared_object(target + ".tmp")
        if found:
            os.replace(target + ".tmp", target + ".rejected")
            raise RuntimeError(f"isa_check: {len(found)} hazard(s) in the built code objects ({n} kernels; the library was left at "
                               f"{target}.rejected, NOT installed):\n  " + "\n  ".join(found[:20]))
    os.replace(target + ".tmp", target)
    global LAST_BUILD_USAGE
    LAST_BUILD_USAGE = usage
    return target


LAST_BUILD_USAGE = {}


class MlpParams(C.Structure):
    _fields_ = [("w", C.c_void_p * 8), ("b", C.c_void_p * 8)] + \
        [(n, C.c_void_p) for n in ("w_feat", "b_feat", "w_alpha", "b_alpha", "w_views", "b_views",
                                   "w_rgb", "b_rgb")]


class RbkParams(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("w_trunk", C.c_void_p * 4), ("b_trunk", C.c_void_p * 4)] + \
        [(n, C.c_void_p) for n in ("w_rb", "b_rb", "w_vb", "b_vb", "w_wb", "b_wb", "w_r", "b_r",
                                   "w_v", "b_v", "w_w", "b_w")]


class MarchCfgC(C.Structure):       # include/lush_march.h: lush_march_cfg
    _fields_ = [("R", C.c_int), ("N_samples", C.c_int), ("N_importance", C.c_int), ("perturb", C.c_float),
                ("raw_noise_std", C.c_float), ("white_bkgd", C.c_int), ("lindisp", C.c_int), ("near_mask", C.c_float),
                ("planes_fwd", C.c_int), ("planes_bwd", C.c_int), ("variant", C.c_int), ("same_net", C.c_int),
                ("packed_coarse", C.c_void_p), ("packed_fine", C.c_void_p), ("packed_bwd_coarse", C.c_void_p),
                ("packed_bwd_fine", C.c_void_p)]


class PackJobC(C.Structure):        # include/lush_march.h: lush_pack_job
    _fields_ = [("net", C.c_int), ("planes", C.c_int), ("variant", C.c_int), ("prm", C.POINTER(MlpParams)), ("packed", C.c_void_p)]


class MarchDraws(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t_rand", "noise_c", "u", "noise_f")]


class MarchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb", "depth", "acc", "density", "rgb0", "depth0", "acc0", "density0", "z_std")]


class MarchGout(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb", "depth", "acc", "rgb0", "depth0", "acc0")]


VIEW_Z, VIEW_RAW, VIEW_WEIGHTS, VIEW_Z_COARSE, VIEW_STASH_COARSE, VIEW_STASH_FINE, VIEW_LIVE_COUNTS = range(7)

_p, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
_SIGS = {
    "lush_abi_version": ([], _i),
    "lush_zgrid": ([_p, _i, _i, _i, _p, _p, _p], _i),
    "lush_zfixed": ([_p, _i, _i, _i, _i, _p, _p], _i),
    "lush_composite_fwd": ([_p, _p, _p, _i, _i, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _i, _p], _i),
    "lush_composite_bwd": ([_p, _p, _p, _i, _i, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _p], _i),
    "lush_composite_bwd_blocks": ([_i], _i),
    "lush_loss_scale": ([_p, _i, _p, _p], _i),
    "lush_sample_merge": ([_p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p], _i),
    "lush_pack_rays_fwd": ([_p, _i, _i, _f, _f, _f, _f, _p, _p], _i),
    "lush_pack_rays_bwd": ([_p, _i, _i, _f, _f, _p, _p, _p], _i),
    "lush_gen_rays": ([_p, _p, _p, _p, _i, _f, _f, _f, _f, _p, _p], _i),
    "lush_gen_rays_image": ([_p, _i, _i, _f, _f, _f, _f, _p, _p], _i),
    "lush_align_rays": ([_p, _p, _p, _i, _p, _i, _i, _ll, _i, _i, _f, _f, _f, _f, _p, _p, _p], _i),
    "lush_consist_loss_fwd_bwd": ([_p, _p, _i, _i, _f, _p, _p, _p], _i),
    "lush_rbk_mlp_fwd": ([C.POINTER(RbkParams), _i, _i, _f, _p, _p], _i),
    "lush_rbk_mlp_bwd": ([C.POINTER(RbkParams), _i, _i, _f, _p, _p, _i, C.POINTER(RbkParams), _p, _i, _p], _i),
    "lush_rbk_warp_fwd": ([_p, _p, _i, _i, _p, _p, _p, _p], _i),
    "lush_rbk_warp_bwd": ([_p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p], _i),
    "lush_rbk_warp_ndc_fwd": ([_p, _p, _i, _i, _p, _i, _f, _f, _f, _f, _p, _p, _p, _p], _i),
    "lush_rbk_warp_ndc_bwd": ([_p, _p, _i, _i, _p, _i, _f, _f, _p, _p, _p, _p, _i, _p, _i, _p], _i),
    "lush_blur_mix_fwd": ([_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p], _i),
    "lush_blur_mix_bwd": ([_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p], _i),
    "lush_wsum_fwd": ([_p, _p, _i, _i, _i, _p, _p], _i),
    "lush_wsum_bwd": ([_p, _p, _i, _i, _i, _p, _p, _p, _p], _i),
    "lush_tonemap_fwd": ([_p, _p, _i, _i, _p, _p], _i),
    "lush_tonemap_bwd": ([_p, _p, _i, _i, _p, _p, _p, _p], _i),
    "lush_noise_act_fwd": ([_p, _i, _p, _p], _i),
    "lush_noise_act_bwd": ([_p, _i, _p, _p, _p], _i),
    "lush_loss_fwd_bwd": ([_p, _p, _p, _i, _f, _p, _p, _p, _p, _p], _i),
    "lush_draws": ([C.c_ulonglong, C.c_ulonglong, _p, _ll, _p, _ll, _p, _ll, _p, _ll, _p], _i),
    "lush_mlp_packed_bytes": ([_i, _i], _sz),
    "lush_pack_plan_bytes": ([_i], _sz),
    "lush_pack_plan_build": ([_p, _i, _p, _sz, _p], _i),
    "lush_pack_plan_run": ([_p, _i, _p, _ll, _p], _i),
    "lush_mlp_pack": ([_i, _i, C.POINTER(MlpParams), _p, _p], _i),
    "lush_mlp_pack_for": ([_i, _i, C.POINTER(MlpParams), _p, _i, _p], _i),
    "lush_mlp_stash_bytes": ([_i, _i, _i, _ll], _sz),
    "lush_mlp_dstash_bytes": ([_i, _i, _ll], _sz),
    "lush_mlp_fwd": ([_i, _i, _i, _p, _p, _i, _i, _p, C.POINTER(MlpParams), _p, _p, _i, _p], _i),
    "lush_mlp_bwd": ([_i, _i, _i, _p, _p, _i, _i, _p, C.POINTER(MlpParams), _p, _p, _p,
                      C.POINTER(MlpParams), _p, _i, _p], _i),
    "lush_mlp_bwd_chain": ([_i, _i, _i, _p, _p, _i, _i, _p, C.POINTER(MlpParams), _p, _p, _p, _p, _i, _p], _i),
    "lush_mlp_bwd_weights": ([_i, _i, _i, _i, _i, C.POINTER(MlpParams), _p, _p, _p, C.POINTER(MlpParams), _i, _p], _i),
    "lush_ray_grad_reduce": ([_p, _p, _i, _i, _p, _p], _i),
    "lush_live_aux_bytes": ([_ll], _sz),
    "lush_live_compact": ([_p, _i, _i, _p, _p, _p, _p, _p, _p], _i),
    "lush_mlp_fwd_live": ([_i, _i, _i, _p, _p, _i, _i, _p, C.POINTER(MlpParams), _p, _p, _p, _i, _p], _i),
    "lush_mlp_bwd_chain_live": ([_i, _i, _i, _p, _p, _i, _i, _p, C.POINTER(MlpParams), _p, _p, _p, _p, _p, _p, _i, _p], _i),
    "lush_mlp_bwd_weights_live": ([_i, _i, _i, _i, _i, C.POINTER(MlpParams), _p, _p, _p, C.POINTER(MlpParams), _p, _i, _p], _i),
    "lush_ray_grad_reduce_live": ([_p, _p, _p, _p, _i, _p, _p], _i),
    "lush_march_workspace_bytes": ([C.POINTER(MarchCfgC)], _sz),
    "lush_march_view": ([C.POINTER(MarchCfgC), _i, C.POINTER(_sz), C.POINTER(_sz)], _i),
    "lush_march_fwd": ([C.POINTER(MarchCfgC), _p, C.POINTER(MlpParams), C.POINTER(MlpParams), C.POINTER(MarchDraws),
                        C.POINTER(MarchOut), _p, _p, _p], _i),
    "lush_march_bwd": ([C.POINTER(MarchCfgC), _p, C.POINTER(MlpParams), C.POINTER(MlpParams), C.POINTER(MarchDraws),
                        C.POINTER(MarchGout), _p, C.POINTER(MlpParams), C.POINTER(MlpParams), _p, _p], _i),
    "lush_adam": ([_p, _p, _p, _p, _ll, _f, _f, _f, _f, _i, _f, _p], _i),
    "lush_step_state_bytes": ([], _sz),
    "lush_step_state_init": ([_p, C.c_ulonglong, _i, C.POINTER(_i), C.c_double, C.c_double, C.c_double, C.c_double, _p], _i),
    "lush_step_state_advance": ([_p, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double, _p], _i),
    "lush_draws_state": ([C.c_ulonglong, C.c_ulonglong, _p, _p, _ll, _p, _ll, _p, _ll, _p, _ll, _p], _i),
    "lush_adam_state": ([_p, _p, _p, _p, _ll, _p, _i, _f, _f, _f, _f, _p], _i),
    "lush_adam_multi": ([_p, _p, _p, _p, _ll, _ll, _ll, _i, _f, _f, _f, _f, _p, _f, _p], _i),
    "lush_adam_state_multi": ([_p, _p, _p, _p, _ll, _ll, _ll, _p, _i, _f, _f, _f, _f, _p], _i),
    "lush_debug_stash_layout": ([_i, _i, _ll, C.POINTER(_ll)], _i),
}
EXPORTS = ["lush_last_error"] + list(_SIGS)
ABI_VERSION = 10
PLANES_F16 = 17          # include/lush_march.h: plane code of ONE fp16 plane (1..3 = bf16 planes)
# include/lush_march.h: LUSH_VARIANT_* (kernel-variant bits of the MLP entry points; 0 = the product's choice)
VARIANT_FWD_HALF, VARIANT_FWD_512, VARIANT_BWD_512, VARIANT_HEAD_KERNEL, VARIANT_BWD_HALF, VARIANT_PE_ROWS, VARIANT_DW_SPLIT, VARIANT_DENSE_BWD = 1, 2, 4, 8, 16, 64, 128, 256
# include/lush_march.h: LUSH_FAULT_*
FAULT_NAMES = {1: "rgb_map", 2: "depth_map", 4: "acc_map", 8: "density_map", 16: "raw", 32: "rgb0", 64: "depth0",
               128: "acc0", 256: "density0", 512: "raw0", 1024: "z_std"}
FAULT_BITS = {n: b for b, n in FAULT_NAMES.items()}
FAULT_COARSE_SHIFT = 5


def fault_names(word: int):
    return [n for b, n in FAULT_NAMES.items() if word & b]


def load():
    """Load the library (building is NOT implicit: call build() first)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be in the process before our library is
    # dlopen'ed, or the library binds a second HIP runtime that sees no device.
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is required; there is no fallback path)")
    lib = C.CDLL(SO_PATH)
    lib.lush_last_error.argtypes = []
    lib.lush_last_error.restype = C.c_char_p
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def call(name: str, *args):
    """Call an int-returning entry point; raise with lush_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.lush_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def mlp_struct(tensors: Sequence, n_layers: int) -> MlpParams:
    """tensors = [w0,b0,...,w{n-1},b{n-1}, views_w, views_b, feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b]."""
    s = MlpParams()
    for l in range(n_layers):
        s.w[l] = tensors[2 * l].data_ptr()
        s.b[l] = tensors[2 * l + 1].data_ptr()
    o = 2 * n_layers
    s.w_views, s.b_views = tensors[o].data_ptr(), tensors[o + 1].data_ptr()
    s.w_feat, s.b_feat = tensors[o + 2].data_ptr(), tensors[o + 3].data_ptr()
    s.w_alpha, s.b_alpha = tensors[o + 4].data_ptr(), tensors[o + 5].data_ptr()
    s.w_rgb, s.b_rgb = tensors[o + 6].data_ptr(), tensors[o + 7].data_ptr()
    return s


def rbk_struct(tensors: Sequence) -> RbkParams:
    """tensors = [embed, (w,b)x4 trunk, r_branch w,b, v_branch w,b, w_branch w,b, r_linear w,b,
    v_linear w,b, w_linear w,b]."""
    s = RbkParams()
    s.embed = tensors[0].data_ptr()
    for l in range(4):
        s.w_trunk[l] = tensors[1 + 2 * l].data_ptr()
        s.b_trunk[l] = tensors[2 + 2 * l].data_ptr()
    names = ("w_rb", "b_rb", "w_vb", "b_vb", "w_wb", "b_wb", "w_r", "b_r", "w_v", "b_v", "w_w", "b_w")
    for k, n in enumerate(names):
        setattr(s, n, tensors[9 + k].data_ptr())
    return s
