"""Host-side mirror of the reference's model interface for the ray-march path.

Class, method and parameter names follow models/lushnerf.py and
utils/run_lushnerf_helpers.py so that (a) a reference state_dict loads here and
vice versa (same 108 keys incl. the triple-aliased RBK) and (b) the
reference's train() can call ``nerf(H, W, K, chunk=..., rays=..., rays_info=..., ...)``
unchanged.  The nn.Linear / nn.Embedding modules are parameter containers only:
all arithmetic runs in the HIP kernels behind lush_nerf_amd.ops.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import MarchCfg, Precision


def _linear_stack(dims):
    return nn.ModuleList([nn.Linear(i, o) for i, o in dims])


class NeRF(nn.Module):
    """Parameter layout of utils/run_lushnerf_helpers.py:365-392 (use_viewdirs=True)."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                 use_viewdirs=True, use_awp=False):
        super().__init__()
        if not use_viewdirs:
            raise NotImplementedError("the MI355X path implements the use_viewdirs=True topology only")
        if use_awp:
            raise NotImplementedError("use_awp=True is not built (utils/run_lushnerf_helpers.py:366-377; no shipped config sets it)")
        if tuple(skips) != (4,) or (D, W) not in ((8, 256), (4, 128)) or input_ch != 63 or input_ch_views != 27:
            raise NotImplementedError("kernels are built for D=8/W=256 (NeRF) and D=4/W=128 (NeRF_Noise), "
                                      "multires=10, multires_views=4, skips=[4]")
        self.D, self.W = D, W
        self.pts_linears = _linear_stack(
            [(input_ch, W)] + [(W + input_ch, W) if (i in skips) else (W, W) for i in range(D - 1)])
        self.views_linears = _linear_stack([(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def tensors(self) -> List[torch.Tensor]:
        t = []
        for l in self.pts_linears:
            t += [l.weight, l.bias]
        for m in (self.views_linears[0], self.feature_linear, self.alpha_linear, self.rgb_linear):
            t += [m.weight, m.bias]
        return t


class NeRF_Noise(NeRF):
    """utils/run_lushnerf_helpers.py:456-512: same layout; alpha_linear is never used."""


class View_Embedding(nn.Module):
    def __init__(self, num_embed, embed_dim):
        super().__init__()
        self.num_embed, self.embed_dim = num_embed, embed_dim
        self.view_embed_layer = nn.Embedding(num_embed, embed_dim)


class Rigid_Blurring_Kernel(nn.Module):
    """models/lushnerf.py:37-153 (parameters + the two public methods)."""

    def __init__(self, D, W, D_r, W_r, D_v, W_v, D_w, W_w, output_ch_r, output_ch_v, input_ch, skips,
                 rv_window, num_motion=2, near=0.0, far=1.0, ndc=False, warp_field=None,
                 view_embedding_layer=None, use_origin=True):
        super().__init__()
        if (D, W, D_r, W_r, D_v, W_v, D_w, W_w, output_ch_r, output_ch_v, input_ch) != \
                (4, 64, 1, 32, 1, 32, 1, 32, 3, 3, 64) or not use_origin or not (1 <= num_motion <= 4):
            raise NotImplementedError("RBK kernels are built for the shipped config (configs/*_lushnerf:36-54)")
        self.num_motion, self.rv_window, self.use_origin = num_motion, rv_window, use_origin
        self.view_embedding_layer = view_embedding_layer
        self.view_embed_linears = _linear_stack([(input_ch, W)] + [(W, W)] * (D - 1))
        self.r_branch = _linear_stack([(W, W_r)])
        self.r_linear = nn.Linear(W_r, output_ch_r * num_motion)
        self.v_branch = _linear_stack([(W, W_v)])
        self.v_linear = nn.Linear(W_v, output_ch_v * num_motion)
        self.w_branch = _linear_stack([(W, W_w)])
        self.w_linear = nn.Linear(W_w, num_motion + 1)
        bound = 1e-5 * 6.0 / (W_r + output_ch_r * num_motion)      # xavier_uniform, reference gain (:62-68)
        for lin in (self.r_linear, self.v_linear):
            nn.init.uniform_(lin.weight, -bound, bound)

    def tensors(self) -> List[torch.Tensor]:
        t = [self.view_embedding_layer.view_embed_layer.weight]
        for l in self.view_embed_linears:
            t += [l.weight, l.bias]
        for m in (self.r_branch[0], self.v_branch[0], self.w_branch[0], self.r_linear, self.v_linear, self.w_linear):
            t += [m.weight, m.bias]
        return t

    def forward(self, rays, rays_info, grad_mask=None, hooks=None):
        """rays [N,3,2], rays_info['images_idx'] [N,1] -> (new_rays [N*(M+1),3,2], ccw [N,M+1])."""
        return ops.RbkWarp.apply(rays, rays_info['images_idx'], self.num_motion, self.rv_window, grad_mask, hooks,
                                 *self.tensors())

    def warp_ndc(self, rays, rays_info, H, W, focal, ndc, near, far, grad_mask=None, hooks=None):
        """forward() followed by the ray prologue of render_train_scene / render_train_noise in one kernel (ops.RbkWarpNdc):
        -> (ray batch [N*(M+1),11] of the warped rays, ccw [N,M+1], ray batch [N,11] of the input rays)."""
        return ops.RbkWarpNdc.apply(rays, rays_info['images_idx'], self.num_motion, self.rv_window, grad_mask, hooks,
                                    H, W, focal, ndc, near, far, *self.tensors())

    def rbk_weighted_sum(self, rgb, depth, acc, extras, ccw):
        rgb, depth, acc = ops.WSum.apply(rgb, ccw), ops.WSum.apply(depth, ccw), ops.WSum.apply(acc, ccw)
        for k, v in extras.items():
            if 1 <= v.dim() <= 3:
                extras[k] = ops.WSum.apply(v, ccw)
        return rgb, depth, acc, extras


class RBK(nn.Module):
    """models/lushnerf.py:156-175."""

    def __init__(self, num_img, view_embed_ch, D_rbk, W_rbk, D_rbk_r, W_rbk_r, D_rbk_v, W_rbk_v, D_rbk_w, W_rbk_w,
                 output_ch_rbk_r, output_ch_rbk_v, skips_rbk, rbk_use_origin, rbk_se_rv_window, num_motion_rbk,
                 use_dpnerf=True, use_awp=False, near=0.0, far=1.0, ndc=False):
        super().__init__()
        if use_awp or not use_dpnerf:
            raise NotImplementedError("RBK: use_dpnerf=True, use_awp=False only (models/lushnerf.py:156-175; configs/*_lushnerf)")
        self.use_dpnerf, self.view_embed_ch, self.use_awp = use_dpnerf, view_embed_ch, use_awp
        self.view_embed_layer = View_Embedding(num_embed=num_img, embed_dim=view_embed_ch)
        self.RBK = Rigid_Blurring_Kernel(
            D=D_rbk, W=W_rbk, num_motion=num_motion_rbk, D_r=D_rbk_r, W_r=W_rbk_r, output_ch_r=output_ch_rbk_r,
            D_v=D_rbk_v, W_v=W_rbk_v, output_ch_v=output_ch_rbk_v, D_w=D_rbk_w, W_w=W_rbk_w,
            rv_window=rbk_se_rv_window, view_embedding_layer=self.view_embed_layer, use_origin=rbk_use_origin,
            input_ch=view_embed_ch, skips=skips_rbk, near=near, far=far, ndc=ndc)


class NeRFAll(nn.Module):
    """models/lushnerf.py:177-989 restricted to the ray-march path."""

    def __init__(self, args, blur_kernel_net=None, precision: Optional[Precision] = None):
        super().__init__()
        self.args = args
        self.precision = precision or Precision()
        self.blur_model_type = args.blur_model_type
        self.blur_kernel_net = blur_kernel_net
        if args.multires != 10 or args.multires_views != 4 or args.i_embed != 0 or not args.use_viewdirs:
            raise NotImplementedError("kernels are built for multires=10, multires_views=4, use_viewdirs")
        if args.rgb_activate != 'sigmoid' or args.sigma_activate != 'relu':
            raise NotImplementedError("kernels implement rgb_activate=sigmoid, sigma_activate=relu (configs/*)")
        if args.tone_mapping_type not in ('gamma', 'none'):
            raise NotImplementedError("tone mapping 'gamma' and 'none' only (no config uses the learned ones)")
        self.input_ch, self.input_ch_views = 63, 27
        self.mlp_coarse = NeRF(D=args.netdepth, W=args.netwidth)
        self.mlp_noise_coarse = NeRF_Noise(D=args.netdepth // 2, W=args.netwidth // 2)
        self.mlp_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine) if args.N_importance > 0 else None
        if blur_kernel_net is not None and self.blur_model_type == 'dpnerf':
            self.dbk_view_embedding = blur_kernel_net.view_embed_layer
            self.mlp_rbk = blur_kernel_net.RBK
        self.gamma = args.tone_mapping_type == 'gamma'
        # numerical-fault word (include/lush_march.h LUSH_FAULT_*): the kernels OR bits into it, read_faults()
        # fetches and clears it.  Not persistent: the reference state_dict has exactly 108 keys.
        self.register_buffer("_faults", torch.zeros(1, dtype=torch.int32), persistent=False)
        self.rng_stream = 0      # Philox stream of the march draws; the trainer sets it to the data-parallel rank
        # per-model switches of the ops (timer / stash hand-out for tests / gradient sink / draw counter): carried here and
        # handed to every op, nothing process-global (DataParallel worker threads share no state)
        self.hooks = ops.Hooks()

    # ------------------------------------------------------------------ helpers
    def tonemapping(self, x, noise_raw=None):
        return ops.ToneMap.apply(x, noise_raw, self.gamma)

    def read_faults(self, clear: bool = True) -> int:
        """NaN/Inf report of every render call since the last read (one device sync).  The reference tests each
        result key after every chunk and prints (models/lushnerf.py:474-478, 578-582); here the kernels set bits
        of one device word and the caller decides when to look.  Returns the LUSH_FAULT_* bit mask."""
        v = int(self._faults.item())
        if v and clear:
            self._faults.zero_()
        return v

    @staticmethod
    def fault_names(word: int):
        from . import lib
        return [n for b, n in lib.FAULT_NAMES.items() if word & b]

    def _draws(self, R, N_samples, N_importance, perturb, raw_noise_std, device, draws):
        """Random draws in the reference's order and shapes (models/lushnerf.py:515, :322;
        helpers:578) unless given explicitly (parity tests): one device-side Philox launch."""
        if draws is not None:
            return draws
        return ops.march_draws(R, N_samples, N_importance, perturb, raw_noise_std, device, stream_id=self.rng_stream, hooks=self.hooks)

    def _march(self, ray_batch, N_samples, N_importance, perturb, raw_noise_std, white_bkgd, lindisp, retraw,
               draws=None):
        near_mask = -1.0
        if (not self.training) and getattr(self.args, "render_rmnearplane", 0) > 0:
            near_mask = self.args.render_rmnearplane / 128
        cfg = MarchCfg(N_samples=N_samples, N_importance=N_importance, perturb=float(perturb),
                       raw_noise_std=float(raw_noise_std), white_bkgd=bool(white_bkgd), lindisp=bool(lindisp),
                       near_mask=near_mask, precision=self.precision, has_fine=self.mlp_fine is not None,
                       want_grad=torch.is_grad_enabled(),
                       flags=self._faults if self._faults.is_cuda else None, hooks=self.hooks)
        R = ray_batch.shape[0]
        d = self._draws(R, N_samples, N_importance, perturb, raw_noise_std, ray_batch.device, draws)
        coarse = self.mlp_coarse.tensors()
        fine = self.mlp_fine.tensors() if (self.mlp_fine is not None and N_importance > 0) else []
        out = ops.March.apply(ray_batch, cfg, d, len(coarse), *coarse, *fine)
        ret = {'rgb_map': out[0], 'depth_map': out[1], 'acc_map': out[2], 'density_map': out[3]}
        if retraw:
            ret['raw'] = out[4]
        if N_importance > 0:
            ret.update(rgb0=out[7], depth0=out[8], acc0=out[9], density0=out[10], z_std=out[11])
        return ret

    def _noise(self, ray_batch, N_samples, lindisp):
        return ops.NoiseMlp.apply(ray_batch.detach(), N_samples, 16, bool(lindisp), self.precision.noise(),
                                  torch.is_grad_enabled(), self.hooks,
                                  *self.mlp_noise_coarse.tensors())

    # ------------------------------------------------------------------ render_rays trio
    def render_rays(self, ray_batch, N_samples, img_idx=None, retraw=False, lindisp=False, perturb=0.,
                    N_importance=0, white_bkgd=False, raw_noise_std=0., pytest=False, force_naive=False,
                    inference=False, draws=None):
        """models/lushnerf.py:354-479 -> (ret, ret_noise)."""
        ret = self._march(ray_batch, N_samples, N_importance, perturb, raw_noise_std, white_bkgd, lindisp, retraw,
                          draws)
        noise = self._noise(ray_batch, N_samples, lindisp)
        ret_noise = {'rgb_map': noise}
        if retraw:
            ret_noise['raw'] = 0
        if N_importance > 0:
            ret_noise['rgb0'] = noise
        return ret, ret_noise

    def render_rays_nonoise(self, ray_batch, N_samples, img_idx=None, retraw=False, lindisp=False, perturb=0.,
                            N_importance=0, white_bkgd=False, raw_noise_std=0., pytest=False, force_naive=False,
                            inference=False, draws=None):
        """models/lushnerf.py:481-583 -> ret."""
        return self._march(ray_batch, N_samples, N_importance, perturb, raw_noise_std, white_bkgd, lindisp, retraw,
                           draws)

    def render_rays_noise(self, ray_batch, N_samples, img_idx=None, retraw=False, lindisp=False, perturb=0.,
                          N_importance=0, white_bkgd=False, raw_noise_std=0., pytest=False, force_naive=False,
                          inference=False, draws=None):
        """models/lushnerf.py:585-617 -> {'rgb_map'}."""
        return {'rgb_map': self._noise(ray_batch, N_samples, lindisp)}

    # ------------------------------------------------------------------ the "render()" trio
    @staticmethod
    def _refuse_unbuilt(c2w_staticcam, use_awp):
        """Branches of the render trio that are not built raise instead of being silently skipped: `c2w_staticcam` (rays of a
        second camera with the first one's view directions, models/lushnerf.py:709-713, 775-779, 830-834) and `use_awp` (the
        adaptive-weight branch, :222-260, 760, 817).  No shipped config sets either.  `kernelpixel` and `allkernel` are accepted and
        unused exactly as in the reference, whose three bodies (:679-866) never read them."""
        if c2w_staticcam is not None:
            raise NotImplementedError("c2w_staticcam is not built (models/lushnerf.py:709-713: a visualisation aid no config uses)")
        if use_awp:
            raise NotImplementedError("use_awp=True is not built (models/lushnerf.py:222-260; no shipped config sets it)")

    def _pack(self, H, W, K, rays, ndc, near, far, use_viewdirs):
        if not use_viewdirs:
            raise NotImplementedError("use_viewdirs=False is not built (every config sets it)")
        sh = rays.shape[:-2]
        return ops.PackRays.apply(rays, H, W, float(K[0][0]), ndc, near, far), sh

    @staticmethod
    def _slice_draws(draws, i, j):
        return None if draws is None else {k: v[i:j] for k, v in draws.items()}

    def _chunks(self, fn, batch, chunk, draws):
        if batch.shape[0] <= chunk:         # one chunk: no slice node (its backward is a zero-fill plus a copy of the whole gradient)
            return [fn(batch, draws)]
        outs = []
        for i in range(0, batch.shape[0], chunk):
            outs.append(fn(batch[i:i + chunk], self._slice_draws(draws, i, i + chunk)))
        return outs

    @staticmethod
    def _merge(dicts, sh):
        out = {}
        for k in dicts[0]:
            v = torch.cat([d[k] for d in dicts], 0) if len(dicts) > 1 else dicts[0][k]
            out[k] = v.reshape(list(sh) + list(v.shape[1:]))
        return out

    def render_infer(self, H, W, K, chunk, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
                     c2w_staticcam=None, use_awp=False, allkernel=0, kernelpixel=None, render_noise=True,
                     draws=None, **kwargs):
        """models/lushnerf.py:679-763 -> ([rgb, depth, acc, extras], noise_rgb)."""
        self._refuse_unbuilt(c2w_staticcam, use_awp)
        batch, sh = self._pack(H, W, K, rays, ndc, near, far, use_viewdirs)
        res = self._chunks(lambda b, d: self.render_rays(b, draws=d, **kwargs), batch, chunk, draws)
        all_ret = self._merge([r[0] for r in res], sh)
        noise = torch.cat([r[1]['rgb_map'] for r in res], 0) if len(res) > 1 else res[0][1]['rgb_map']      # (one chunk: no copy launch)
        k_extract = ['rgb_map', 'depth_map', 'acc_map']
        return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}], noise

    def render_train_scene(self, H, W, K, chunk, rays=None, c2w=None, ndc=True, near=0., far=1.,
                           use_viewdirs=False, c2w_staticcam=None, use_awp=False, allkernel=0, kernelpixel=None,
                           render_noise=True, draws=None, **kwargs):
        """models/lushnerf.py:766-819 -> [rgb, depth, acc, extras]."""
        self._refuse_unbuilt(c2w_staticcam, use_awp)
        batch, sh = self._pack(H, W, K, rays, ndc, near, far, use_viewdirs)
        res = self._chunks(lambda b, d: self.render_rays_nonoise(b, draws=d, **kwargs), batch, chunk, draws)
        all_ret = self._merge(res, sh)
        k_extract = ['rgb_map', 'depth_map', 'acc_map']
        return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]

    def _render_train_scene_packed(self, batch, chunk, draws=None, **kwargs):
        """render_train_scene on an already packed ray batch [R,11] (the fused warp + NDC kernel made it)."""
        self._refuse_unbuilt(kwargs.get("c2w_staticcam"), kwargs.get("use_awp", False))
        for k in ("c2w", "ndc", "near", "far", "use_viewdirs", "c2w_staticcam", "use_awp", "allkernel", "kernelpixel", "render_noise"):
            kwargs.pop(k, None)
        res = self._chunks(lambda b, d: self.render_rays_nonoise(b, draws=d, **kwargs), batch, chunk, draws)
        all_ret = self._merge(res, batch.shape[:-1])
        k_extract = ['rgb_map', 'depth_map', 'acc_map']
        return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]

    def _render_train_noise_packed(self, batch0, chunk, **kwargs):
        """render_train_noise on an already packed ray batch [N,11]."""
        res = self._chunks(lambda b, d: self._noise(b, kwargs['N_samples'], kwargs.get('lindisp', False)), batch0, chunk, None)
        return torch.cat(res, 0) if len(res) > 1 else res[0]

    def render_train_noise(self, H, W, K, chunk, rays=None, c2w=None, ndc=True, near=0., far=1.,
                           use_viewdirs=False, c2w_staticcam=None, use_awp=False, allkernel=0, kernelpixel=None,
                           render_noise=True, draws=None, **kwargs):
        """models/lushnerf.py:821-866 -> noise rgb [N,3]."""
        self._refuse_unbuilt(c2w_staticcam, use_awp)
        batch, _ = self._pack(H, W, K, rays, ndc, near, far, use_viewdirs)
        res = self._chunks(lambda b, d: self.render_rays_noise(b, **kwargs)['rgb_map'], batch, chunk, None)
        return torch.cat(res, 0) if len(res) > 1 else res[0]

    # ------------------------------------------------------------------ forward
    def forward(self, H, W, K, chunk=1024 * 32, rays=None, rays_info=None, poses=None, allkernel=False,
                kernel_pixel=None, consist_loss=False, Align_matrix=None, Align_mask=None, anchor_pose=None,
                samples=None, **kwargs):
        """models/lushnerf.py:619-677.  Training branch returns the reference 7-tuple."""
        if self.training and not consist_loss:
            assert rays is not None, "Please specify rays when in the training mode"
            force_baseline = bool(kwargs['force_naive'])
            if self.blur_kernel_net is not None and not force_baseline and self.blur_model_type == 'dpnerf':
                kwargs['img_idx'] = rays_info['images_idx'].squeeze(-1)
                mask = kernel_pixel if allkernel else None      # torch.where(mask, x, x.detach()) (:641-643)
                if not kwargs.get('use_viewdirs', False):
                    raise NotImplementedError("use_viewdirs=False is not built (every config sets it)")
                # (:638-654) warp + ray prologue as ONE kernel, both marches, then the weighted sums, the noise colour and the
                # five tone-mapped outputs as ONE kernel (ops.RbkWarpNdc / ops.BlurMix; the piecewise ops -- mlp_rbk(),
                # render_train_scene / _noise, rbk_weighted_sum, tonemapping -- stay available and are tested against these)
                batch, ccw, batch0 = self.mlp_rbk.warp_ndc(rays, rays_info, H, W, float(K[0][0]), kwargs.get('ndc', True),
                                                            kwargs.get('near', 0.), kwargs.get('far', 1.), grad_mask=mask,
                                                            hooks=self.hooks)
                rgb, depth, acc, extras = self._render_train_scene_packed(batch, chunk, **kwargs)
                noise_raw = self._render_train_noise_packed(batch0, chunk, **kwargs)
                blur, blur0, rgb_noise, sharp, sharp0 = ops.BlurMix.apply(rgb, extras['rgb0'], ccw, noise_raw, self.gamma)
                return blur, blur0, {}, rgb_noise, rgb_noise, sharp, sharp0
            kwargs['img_idx'] = rays_info['images_idx'].squeeze(-1)
            (rgb, depth, acc, extras), noise_raw = self.render_infer(H, W, K, chunk, rays, **kwargs)
            rgb_noise = ops.NoiseAct.apply(noise_raw)
            return self.tonemapping(rgb), self.tonemapping(extras['rgb0']), {}, rgb_noise, rgb_noise, {}, {}
        if self.training and consist_loss:      # models/lushnerf.py:664-668
            kwargs['render_kwargs'].pop('save_warped_ray_img', False)
            return self.Render_Aligned_Pixel(H, W, K, chunk, poses, images_idx=rays_info, align_matrix=Align_matrix,
                                             align_mask=Align_mask, anchor_pose=anchor_pose, samples=samples, **kwargs)
        assert poses is not None, "Please specify poses when in the eval model"
        kwargs['render_kwargs'].pop('save_warped_ray_img', False)
        rgbs, rgbs_noise, depths = self.render_path(H, W, K, chunk, poses, **kwargs)
        rgbs_noise = ops.NoiseAct.apply(torch.reshape(rgbs_noise, [-1, H, W, 3]))
        return self.tonemapping(rgbs), self.tonemapping(rgbs_noise), depths

    def render_path(self, H, W, K, chunk, render_poses, render_kwargs, render_factor=0):
        """models/lushnerf.py:868-896 (eval): one render_infer per pose."""
        if render_factor != 0:
            H, W = H // render_factor, W // render_factor
        rgbs, depths, noises = [], [], []
        for c2w in render_poses:
            rays = ops.gen_rays_image(c2w, H, W, K)       # get_rays (helpers:517-528) on the device, [H,W,3,2]
            with torch.no_grad():
                (rgb, depth, acc, extras), noise = self.render_infer(H, W, K, chunk=chunk, rays=rays,
                                                                     c2w=c2w[:3, :4], **render_kwargs)
            rgbs.append(rgb); depths.append(depth); noises.append(noise)
        return torch.stack(rgbs, 0), torch.stack(noises, 0), torch.stack(depths, 0)

    def Render_Aligned_Pixel(self, H, W, K, chunk, render_poses, images_idx, render_kwargs, render_factor=0,
                             align_matrix=None, align_mask=None, anchor_pose=None, samples=None):
        """models/lushnerf.py:949-989 -> (rgb_align [V,32,3], align_certainty [V,32]).

        The reference draws `anchor_pose = random.randint(0, V-1)` and `samples = np.random.randint(0, 640*1120,
        32)` on the host (:960, :964); pass them explicitly to pin a run (parity tests), otherwise they are
        drawn the same way here.  The reference then loops over the poses, builds each pose's full H*W ray
        table and renders 32 rays at a time; rays are independent and this branch draws nothing (perturb
        False, raw_noise_std 0), so one gather kernel + ONE render_train_scene call over V*32 rays gives the
        same values."""
        if render_factor != 0:
            H, W = H // render_factor, W // render_factor
        V = len(render_poses)
        if anchor_pose is None:
            import random
            anchor_pose = random.randint(0, V - 1)
        if samples is None:
            import numpy as np
            samples = torch.from_numpy(np.random.randint(0, 640 * 1120, size=32))
        samples = torch.as_tensor(samples).long()
        c2w = render_poses if torch.is_tensor(render_poses) else torch.stack(list(render_poses), 0)
        rays, certainty = ops.align_rays(c2w, align_matrix[anchor_pose], align_mask[anchor_pose], samples, H, W, K)
        rgb, _, _, _ = self.render_train_scene(H, W, K, chunk=chunk, rays=rays.detach(), render_noise=False,
                                               **render_kwargs)
        return rgb.reshape(V, samples.numel(), 3), certainty


def load_reference_weights(model: NeRFAll, weights: Dict[str, "torch.Tensor"]):
    """Copy a {canonical reference name -> array} dict (lush_nerf_amd.synth / a reference
    state_dict with the 'module.' prefix stripped) into the model."""
    sd = model.state_dict()
    new = {}
    for k in sd:
        ck = k
        if k.startswith("blur_kernel_net.RBK."):
            ck = "mlp_rbk." + k[len("blur_kernel_net.RBK."):]
        elif k.startswith("blur_kernel_net.view_embed_layer.") or k.startswith("dbk_view_embedding."):
            ck = "mlp_rbk.view_embedding_layer.view_embed_layer.weight"
        src = weights[ck] if ck in weights else weights[k]
        new[k] = torch.as_tensor(src).to(sd[k].dtype)
    model.load_state_dict(new, strict=True)
    return model
