"""Samplers.  ``EulerEDMSampler`` is the one UDiffText instantiates (reference util.py:35-45); with s_churn = 0 it
is the deterministic Euler == DDIM(eta 0) integrator with classifier-free guidance.

Reference: sgm/modules/diffusionmodules/sampling.py — BaseDiffusionSampler :28-78, EDMSampler :89-98,
EulerEDMSampler :218-420 (get_init_noise :264-322, sampler_step :324-353, __call__ :355-420).

MI355X execution of one step (``_Stepper.step``): all scalars (sigma, quantised sigma, timestep index,
c_in, c_out) are computed on the host from the fp32 tables — no ``.item()`` sync in the loop; the device work is
  udt_unet_input (x*c_in into the NHWC bf16 CFG pair) -> UNet kernels -> udt_cfg_euler_step (c_out, CFG, Euler),
with the step-invariant pieces hoisted out of the loop (text k|v projections of all 16 transformers, the
concat channels of the UNet input).  Attention maps are only emitted where they are consumed (noise search).
The heavier side paths (attend-and-excite: needs autograd through the UNet; attention-map plots / GIFs) are
out of scope and raise.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional, Union

import numpy as np
import torch

from udifftext_amd import ops, packing, rng

from ...util import default, instantiate_from_config, require_gpu
from .guiders import VanillaCFG
from .sampling_utils import to_d

DEFAULT_GUIDER = {"target": "sgm.modules.diffusionmodules.guiders.IdentityGuider"}


class BaseDiffusionSampler:
    def __init__(self, discretization_config, num_steps: Union[int, None] = None, guider_config=None,
                 verbose: bool = False, device: str = "cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device=self.device)
        uc = default(uc, cond)
        x *= torch.sqrt(1.0 + sigmas[0] ** 2.0)
        s_in = x.new_ones([x.shape[0]])
        return x, s_in, sigmas, len(sigmas), cond, uc

    def denoise(self, x, model, sigma, cond, uc):
        """generic (any denoiser / network) formulation, tensor math as in the reference :61-64"""
        denoised = model.denoiser(model.model, *self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)

    def get_sigma_gen(self, num_sigmas, init_step=0):
        gen = range(init_step, num_sigmas - 1)
        if self.verbose:
            try:
                from tqdm import tqdm
                gen = tqdm(gen, total=num_sigmas - 1 - init_step,
                           desc=f"Sampling with {self.__class__.__name__} for {num_sigmas - 1 - init_step} steps")
            except ImportError:
                pass
        return gen


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc, *args, **kwargs):
        raise NotImplementedError

    def euler_step(self, x, d, dt):
        return x + dt * d


class EDMSampler(SingleStepDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise


# noise search: candidates as extra batch entries of one UNet call (0: one candidate at a time)
NOISE_BATCH = os.environ.get("UDT_NOISE_BATCH", "1") != "0"
# UDT_DUAL_STREAM=1: the uc / c halves of a LONE batch's UNet call on two streams.  Default OFF since round 4: with the wide
# convolution (one 256-pixel tile per CU for the full 8-sample call) one stream is 1.5 % faster (9.60 vs 9.75 ms per step,
# profiles/r04_ab_dual_stream.txt); rounds 1-3 gained 3-8 % from the split
DUAL_STREAM = os.environ.get("UDT_DUAL_STREAM", "0") != "0"


def _all_zero(t: torch.Tensor) -> bool:
    """is the (unconditional) text context exactly zero?  GeneralConditioner tags force-zeroed embeddings on the host
    (``_udt_all_zero``); anything else is asked on the device (one host sync)"""
    flag = getattr(t, "_udt_all_zero", None)
    return bool(flag) if flag is not None else not bool(t.any())


def weights_fingerprint(model) -> int:
    """changes whenever a parameter of ``model`` is re-assigned, moved or written in place (load_state_dict,
    init_from_ckpt, .to()): captured hipGraphs bake in the device pointers of the packed weights, so the graph caches
    are keyed by it (~1 ms for the engine's 1330 tensors, once per sampling call)"""
    h = 0
    for p in model.parameters():
        h = (h * 1000003 + p._version * 8191 + p.data_ptr()) & 0xFFFFFFFFFFFFFFFF
    return h


def _is_capture_failure(e: BaseException) -> bool:
    """only 'this stream / runtime cannot capture' errors may downgrade the sampler to eager launches"""
    from udifftext_amd import lib as L
    if isinstance(e, (L.UdtError, torch.OutOfMemoryError)):
        return False
    msg = str(e).lower()
    return "captur" in msg or "graph" in msg


class _Stepper:
    """Step-invariant device state of one sampling run + the fused per-step launch sequence."""

    def __init__(self, model, cond: dict, uc: dict, batch_size: int, latent_hw, scale: float, two_streams=None):
        self.engine = model
        self.unet = model.model.diffusion_model
        self.scale = float(scale)
        self.B = batch_size
        h, w = latent_hw
        dev = cond["concat"].device
        self.table = model.denoiser.sigmas.detach().float().cpu()          # ascending 1000-entry table
        ctx = torch.cat((uc["t_crossattn"], cond["t_crossattn"]), 0)
        self.t_kv = self.unet.project_context(ctx)                        # hoisted k|v of all transformers
        self.t_fused = self.unet.prepare_fused_tattn(self.t_kv)           # ... folded further into the fused t_attn tables
        # force_uc_zero_embeddings=["label"] (reference sample loop) makes the unconditional context exactly zero:
        # its cross-attention is then x + to_out.bias — one host sync per sampling run buys half of every t_attn
        self.zero_ctx_rows = batch_size if _all_zero(uc["t_crossattn"]) else 0
        # two launch streams: the unconditional and the conditional half of the CFG pair never meet before the
        # guidance step, so each runs the UNet on its own HIP stream, planned for half of the CUs.  Measured on
        # MI355X: one stream leaves the chip idle during every kernel's ramp-up / epilogue / tail (a half-GPU plan
        # alone is only 24 % slower than the whole-GPU plan); two concurrent streams fill those holes.
        self.dual = DUAL_STREAM if two_streams is None else bool(two_streams)
        if self.dual:
            self.side = torch.cuda.Stream(device=dev)
            self.eps = torch.empty((2 * batch_size, h, w, 4), dtype=torch.float32, device=dev)
            self.t_kv_u = [[kv[:batch_size] for kv in lst] for lst in self.t_kv]
            self.t_kv_c = [[kv[batch_size:] for kv in lst] for lst in self.t_kv]
            self.t_fused_u = [[tb.rows(0, batch_size) if tb is not None else None for tb in lst] for lst in self.t_fused]
            self.t_fused_c = [[tb.rows(batch_size) if tb is not None else None for tb in lst] for lst in self.t_fused]
        self.xin = torch.zeros((2 * batch_size, h, w, packing.KPAD), dtype=torch.bfloat16, device=dev)
        concat = torch.cat((uc["concat"], cond["concat"]), 0).float().contiguous()
        ops.nhwc_set_channels(concat, self.xin, 4)                         # channels 4..8: mask, masked latent
        self._emb_cache: Dict[int, torch.Tensor] = {}
        self.dev = dev
        # stream-K workspaces owned by this stepper (one per launch stream): allocated with it, referenced by the
        # graphs captured from it, freed with it
        self.ws = ops.Workspace(dev)
        self.ws_side = ops.Workspace(dev) if self.dual else None
        # launch streams of OTHER runners sharing the device: the caller's ambient share (a lane of predict_many runs its
        # conditioning AND its noise search under launch_context(cu_share=n)); _GraphedSteps overrides it for its runner
        self.cu_share = max(1, int(ops._ctx.cu_share))

    def quantise(self, sigma: float):
        idx = int((self.table - sigma).abs().argmin())
        return idx, float(self.table[idx])

    def emb_rows(self, idx: int) -> torch.Tensor:
        rows = self._emb_cache.get(idx)
        if rows is None:
            t = torch.full((2 * self.B,), float(idx), dtype=torch.float32, device=self.dev)
            rows = self.unet.time_embedding_rows(t)
            self._emb_cache[idx] = rows
        return rows

    def step(self, x: torch.Tensor, sigma: float, sigma_next: float, emit_maps: bool = False,
             denoised: Optional[torch.Tensor] = None) -> None:
        """in-place Euler update of x (fp32 NCHW [B,4,h,w]); denoised (optional): receives the guided denoised latent"""
        idx, sq = self.quantise(sigma)
        c_in = 1.0 / (sq * sq + 1.0) ** 0.5
        ops.unet_input(x, self.xin, c_in)
        if emit_maps:
            self.unet.clear_attn_map()
        emb = self.emb_rows(idx)
        if self.dual and not emit_maps:
            eps = self._forward_two_streams(emb)
        else:
            with ops.launch_context(cu_share=self.cu_share, workspace=self.ws):
                eps = self.unet.forward_nhwc(self.xin, emb, self.t_kv, emit_maps=emit_maps,
                                             zero_ctx_rows=self.zero_ctx_rows, t_fused=self.t_fused)
        ops.cfg_euler_step(x, eps, sigma, sigma_next, self.scale, denoised=denoised, c_out=-sq)

    def check(self) -> None:
        """synchronise and raise if a stream-K launch of this stepper timed out (library err word)"""
        self.ws.check()
        if self.ws_side is not None:
            self.ws_side.check(self.side.cuda_stream)

    def _forward_two_streams(self, emb: torch.Tensor) -> torch.Tensor:
        """uc half on the current stream, c half on the side stream (fork / join; captured as two graph branches);
        the stream-K kernels of both streams must be co-resident: each is planned for half of this stepper's CUs"""
        B = self.B
        main = torch.cuda.current_stream()
        share = 2 * self.cu_share
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side), ops.launch_context(cu_share=share, workspace=self.ws_side):
            eps_c = self.unet.forward_nhwc(self.xin[B:], emb[B:], self.t_kv_c, zero_ctx_rows=0, t_fused=self.t_fused_c)
            self.eps[B:].copy_(eps_c)
        with ops.launch_context(cu_share=share, workspace=self.ws):
            eps_u = self.unet.forward_nhwc(self.xin[:B], emb[:B], self.t_kv_u, zero_ctx_rows=self.zero_ctx_rows,
                                           t_fused=self.t_fused_u)
            self.eps[:B].copy_(eps_u)
        main.wait_stream(self.side)
        return self.eps


class _GraphedSteps:
    """hipGraph replay of the sampling loop.  One sampler step is ~580 kernel launches whose arguments depend only on
    (step index, shapes); each step's launch sequence is captured once into a hipGraph on a private memory pool and
    replayed for every later batch of the same shape: the host then issues 50 graph launches per batch instead of
    ~29,000 kernel launches (8 ms of Python / ctypes time per step), and the command processor walks the kernels
    back to back.  Conditioning (text k|v projections, concat channels) lives in static device buffers that
    ``rebind`` refreshes in place; the latent is a static fp32 buffer."""

    def __init__(self, model, cond, uc, batch_size, latent_hw, scale, sig, cu_share: int = 1):
        # cu_share > 1: this runner is one of several batches in flight; its launches are planned for 1/cu_share of
        # the CUs and its UNet stays on one stream (the concurrency comes from the other batches)
        self.cu_share = int(cu_share)
        self.st = _Stepper(model, cond, uc, batch_size, latent_hw, scale, two_streams=None if cu_share == 1 else False)
        self.st.cu_share = self.cu_share
        self.fingerprint = weights_fingerprint(model)
        h, w = latent_hw
        self.x = torch.zeros((batch_size, 4, h, w), dtype=torch.float32, device=self.st.dev)
        self.sig = list(sig)
        self.graphs: Dict[int, torch.cuda.CUDAGraph] = {}
        self.pool = torch.cuda.graph_pool_handle()
        self.capture_stream = torch.cuda.Stream(device=self.st.dev)
        self.warm = False

    def rebind(self, cond, uc) -> bool:
        """refresh the static conditioning buffers for a new batch; False if the launch sequence would differ"""
        st = self.st
        zero_rows = st.B if _all_zero(uc["t_crossattn"]) else 0
        if zero_rows != st.zero_ctx_rows or cond["concat"].shape[0] != st.B:
            return False
        ctx = torch.cat((uc["t_crossattn"], cond["t_crossattn"]), 0)
        for dst_list, src_list in zip(st.t_kv, st.unet.project_context(ctx)):
            for dst, src in zip(dst_list, src_list):
                dst.copy_(src)
        st.unet.prepare_fused_tattn(st.t_kv, out=st.t_fused)              # tables refreshed in place (views stay valid)
        concat = torch.cat((uc["concat"], cond["concat"]), 0).float().contiguous()
        ops.nhwc_set_channels(concat, st.xin, 4)
        return True

    def _capture(self, i: int) -> torch.cuda.CUDAGraph:
        st = self.st
        st.emb_rows(st.quantise(self.sig[i])[0])                 # time-embedding rows are cached outside the graph
        if not self.warm:
            # one eager pass on the capture stream: sets kernel attributes, allocates the library's pages
            torch.cuda.synchronize()
            with torch.cuda.stream(self.capture_stream):
                keep = self.x.clone()
                st.step(self.x, self.sig[i], self.sig[i + 1])
                self.x.copy_(keep)
            torch.cuda.synchronize()
            self.warm = True
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (the RCCL watchdog under torch.distributed) may touch the
        # HIP runtime while this thread captures
        with torch.cuda.graph(g, pool=self.pool, stream=self.capture_stream, capture_error_mode="thread_local"):
            st.step(self.x, self.sig[i], self.sig[i + 1])
        self.graphs[i] = g
        return g

    def run(self, x: torch.Tensor, steps) -> torch.Tensor:
        self.x.copy_(x)
        for i in steps:
            g = self.graphs.get(i)
            if g is None:
                g = self._capture(i)
            g.replay()
        out = self.x.clone()
        self.st.check()                       # one sync per sampling loop: surfaces a stream-K time-out as UdtError
        return out


AAE_GRAPH = os.environ.get("UDT_AAE_GRAPH", "1") != "0"      # hipGraph replay of the attend-and-excite gradient (A/B switch)


class EulerEDMSampler(EDMSampler):
    use_graphs = os.environ.get("UDT_GRAPHS", "1") != "0"      # hipGraph replay of the main loop (eager launches if off)

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step

    # ----------------------------------------------------------------------------------------- helpers
    def _host_sigmas(self, num_steps=None):
        n = self.num_steps if num_steps is None else num_steps
        return [float(s) for s in self.discretization(n, device="cpu")]

    def _check_fast_path(self):
        if not isinstance(self.guider, VanillaCFG):
            raise NotImplementedError("the fused MI355X step implements VanillaCFG guidance")
        if self.s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic sampling) is not used by UDiffText (util.py:39)")

    # -------------------------------------------------------------------------------------- noise search
    def get_init_noise(self, cfgs, model, cond, batch, uc=None):
        """noise_iters candidates, each scored by the text-attention local loss after the 2nd of 2 Euler steps;
        the per-sample arg-min is kept (identical to the reference for batch 1; the reference is undefined for
        larger batches).  All randn draws come from the CPU default generator, in the reference's order."""
        self._check_fast_path()
        H, W = batch["target_size_as_tuple"][0]
        shape = (cfgs.batch_size, cfgs.channel, int(H) // cfgs.factor, int(W) // cfgs.factor)
        dev = cond["concat"].device
        randn = rng.randn_on(shape, dev)
        if cfgs.noise_iters <= 0:
            return randn
        sig = self._host_sigmas(2)
        mask, seg = batch["mask"], batch["seg_mask"]
        B, K = shape[0], int(cfgs.noise_iters)
        uc = default(uc, cond)
        # the reference draws the first candidate, then one more after scoring each (the last draw is never used but advances the
        # generator): K + 1 draws in the same order
        cands = [randn] + [rng.randn_on(shape, dev) for _ in range(K)]
        cands, scores = cands[:K], []
        # candidates are independent of each other (2 Euler steps + the local loss of THAT candidate's attention maps), so they
        # run as extra batch entries of the same UNet calls: up to 16 samples (32 with the CFG pair) per call instead of K
        # sequential 2-step runs on B samples — the reference-default workload (batch 1, noise_iters 10) is launch-latency-bound
        # at 2 samples per call.  UDT_NOISE_BATCH=0: one candidate at a time (A/B, and the regime of the round-2 numbers)
        G = max(1, min(K, 16 // max(1, B))) if NOISE_BATCH else 1
        tile = lambda d, g: {k: (v.repeat((g,) + (1,) * (v.dim() - 1)) if torch.is_tensor(v) else v) for k, v in d.items()}
        steppers = {}
        for g0 in range(0, K, G):
            chunk = cands[g0:g0 + G]
            g = len(chunk)
            stepper = steppers.get(g)
            if stepper is None:
                stepper = steppers[g] = _Stepper(model, tile(cond, g), tile(uc, g), g * B, shape[2:], self.guider.scale)
            x = torch.cat(chunk, 0).clone()
            x *= (1.0 + sig[0] ** 2.0) ** 0.5
            ll = None
            for i in range(2):
                stepper.step(x, sig[i], sig[i + 1], emit_maps=True)
                ll = model.loss_fn.get_min_local_loss(stepper.unet.attn_map_cache, mask, seg, cond_only=True)
            scores.extend(ll.reshape(g, B).unbind(0))
            stepper.unet.clear_attn_map()
            stepper.check()
        score = torch.stack(scores, 0)                                   # [iters, B]
        best = score.argmin(dim=0)                                        # first minimum, like the stable sort
        print(f"Init local loss: Best {score.min().item()} Worst {score.max().item()}")
        stack = torch.stack(cands, 0)                                     # [iters, B, 4, h, w]
        return stack[best, torch.arange(shape[0], device=dev)].contiguous()

    # --------------------------------------------------------------------------------- attend-and-excite
    def get_c_noise(self, x, model, sigma):
        """reference sampling.py:224-231: the quantised timestep index of sigma (EpsScaling: c_noise = sigma)"""
        sigma = model.denoiser.possibly_quantize_sigma(sigma)
        return model.denoiser.possibly_quantize_c_noise(sigma.reshape(-1))

    def attend_and_excite(self, x, model, sigma, cond, batch, alpha, iter_enabled, thres, max_iter=20):
        """reference sampling.py:233-252: x <- x - alpha * d local_loss / d x, once, or (iter_enabled) until the loss falls to
        ``thres`` or max_iter is passed.  The network sees the RAW x (no c_in scaling — the reference calls model.model directly)
        and only the conditional batch; the gradient runs through the HIP path's written-out reverse pass
        (udifftext_amd.backward.unet_local_loss_grad) — per sample, where the reference only accepts B = 1."""
        from udifftext_amd import backward
        require_gpu(x, "EulerEDMSampler.attend_and_excite")
        c_noise = self.get_c_noise(x, model, sigma)
        unet = model.model.diffusion_model
        x = x.detach().clone().float().contiguous()
        args = (c_noise.float(), cond["concat"], cond["t_crossattn"], batch["mask"], batch["seg_mask"])
        evaluate = lambda xx: backward.unet_local_loss_grad(unet, model.loss_fn, xx, *args)
        if self.use_graphs and AAE_GRAPH:
            # the evaluation's launch sequence depends on the shapes only: captured once per sampler, replayed for every update of
            # every step (the timestep index, latent and conditioning are device data in static buffers)
            runner = getattr(self, "_aae_runner", None)
            if runner is None or not runner.valid_for(unet, x, *args):
                runner = self._aae_runner = None                           # (drop the old pool before the new capture)
                try:
                    runner = backward.GraphedLocalLossGrad(unet, model.loss_fn, x, *args)
                    runner(x, *args)
                    self._aae_runner = runner
                except Exception as e:
                    if not _is_capture_failure(e):
                        raise
                    print(f"[udifftext_amd] attend-and-excite: hipGraph capture unavailable ({type(e).__name__}: {e}); eager launches")
                    runner = None
            if runner is not None:
                evaluate = lambda xx: runner(xx, *args)
        iters = 0
        while True:
            loss, grad = evaluate(x)
            ops.axpy_(x, grad, -float(alpha))
            iters += 1
            if not iter_enabled or bool((loss <= thres).all()) or iters > max_iter:
                break
        self.aae_evaluations = getattr(self, "aae_evaluations", 0) + iters     # (how many gradients a sampling run took)
        return x

    # ------------------------------------------------------------------------------------------- API step
    def sampler_step(self, sigma, next_sigma, model, x, cond, batch=None, uc=None, gamma=0.0, alpha=0, iter_enabled=False,
                     thres=None, update=False, name=None, save_loss=False, save_attn=False, save_inter=False):
        """reference-shaped single step on tensors (sigma / next_sigma are [B] tensors); returns
        (x_next, denoised_decode, local_loss).  Generic formulation via denoiser + guider."""
        if gamma > 0:
            raise NotImplementedError("s_churn > 0 (stochastic sampling) is not used by UDiffText (util.py:39)")
        if update:
            x = self.attend_and_excite(x, model, sigma, cond, batch, alpha, iter_enabled, thres)
        denoised = self.denoise(x, model, sigma, cond, uc)
        inter = model.decode_first_stage(denoised) if save_inter else None
        if save_loss:
            ll = model.loss_fn.get_min_local_loss(model.model.diffusion_model.attn_map_cache, batch["mask"], batch["seg_mask"])
            ll = ll[ll.shape[0] // 2:]
        else:
            ll = torch.zeros(1)
        if save_attn:                                                      # reference sampling.py:344-346
            attn_map = model.model.diffusion_model.save_attn_map(save_name=name, tokens=batch["label"][0])
            self.save_segment_map(attn_map, tokens=batch["label"][0], save_name=name)
        d = to_d(x, sigma, denoised)
        dt = (next_sigma - sigma)[(...,) + (None,) * (x.ndim - 1)]
        return self.euler_step(x, d, dt), inter, ll

    def save_segment_map(self, attn_maps, tokens=None, save_name=None, out_dir="./temp/seg_map"):
        """reference sampling.py:254-262: the per-token heat maps of save_attn_map for the label's characters -> one .npy"""
        import numpy as np
        section = np.stack([attn_maps[i] for i in range(len(tokens))])
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, f"seg_{save_name}.npy"), section)
        return section

    # --------------------------------------------------------------------------------------------- loop
    def __call__(self, model, x, cond, batch=None, uc=None, num_steps=None, init_step=0, name=None, aae_enabled=False,
                 detailed=False):
        self._check_fast_path()
        require_gpu(x, "EulerEDMSampler")
        uc = default(uc, cond)
        if aae_enabled:
            return self._sample_with_attend_and_excite(model, x, cond, batch, uc, num_steps, init_step, name, detailed)
        sig = self._host_sigmas(num_steps)
        x = x.float().contiguous()
        x *= (1.0 + sig[0] ** 2.0) ** 0.5                                  # in place, like the reference :54
        if detailed:
            # reference sampling.py:384,344-346: at the middle step the text cross-attention maps of the configured layers are
            # plotted and the label's per-character maps saved.  That one step runs with map emission (eager launches, the xattn
            # chain); every other step is the fast step — the loop is the same Euler loop, so the latent equals the plain call's
            # up to the two text-attention forms' rounding
            name = name if name is not None else (batch["name"][0] if batch is not None and "name" in batch else "sample")
            stepper = _Stepper(model, cond, uc, x.shape[0], x.shape[2:], self.guider.scale)
            mid = (len(sig) - 1) // 2
            for i in self.get_sigma_gen(len(sig), init_step=init_step):
                stepper.step(x, sig[i], sig[i + 1], emit_maps=(i == mid))
                if i == mid:
                    attn_map = stepper.unet.save_attn_map(save_name=name, tokens=batch["label"][0])
                    self.save_segment_map(attn_map, tokens=batch["label"][0], save_name=name)
            stepper.check()
            return x
        if self.use_graphs:
            out = self._run_graphed(model, x, cond, uc, sig, init_step)
            if out is not None:
                return out
        stepper = _Stepper(model, cond, uc, x.shape[0], x.shape[2:], self.guider.scale)
        prev = stepper.unet.cache_attn_maps
        for i in self.get_sigma_gen(len(sig), init_step=init_step):
            stepper.step(x, sig[i], sig[i + 1], emit_maps=False)
        stepper.unet.cache_attn_maps = prev
        stepper.check()
        return x

    def _sample_with_attend_and_excite(self, model, x, cond, batch, uc, num_steps, init_step, name, detailed):
        """reference sampling.py:355-420 with aae_enabled: before every denoising step the latent takes attend-and-excite updates
        (alpha = 20 sqrt(scale_i); iterated at steps 5, 9, ..., 25 down to thresholds -0.5 ... -0.8), the local loss of every step
        is collected and every intermediate denoised latent decoded (the reference writes them as a GIF; imageio is optional
        here).  Eager launches: the update's step count is data-dependent."""
        import numpy as np
        sig = self._host_sigmas(num_steps)
        num_sigmas = len(sig)
        x = x.float().contiguous()
        x *= (1.0 + sig[0] ** 2.0) ** 0.5
        name = name if name is not None else (batch["name"][0] if batch is not None and "name" in batch else "sample")
        scales = np.linspace(start=1.0, stop=0, num=num_sigmas)
        iter_lst = np.linspace(start=5, stop=25, num=6, dtype=np.int32)
        thres_lst = np.linspace(start=-0.5, stop=-0.8, num=6)
        B = x.shape[0]
        stepper = _Stepper(model, cond, uc, B, x.shape[2:], self.guider.scale)
        s_in = x.new_ones([B])
        evals0 = getattr(self, "aae_evaluations", 0)
        inters, local_losses = [], []
        mid = (num_sigmas - 1) // 2
        for i in self.get_sigma_gen(num_sigmas, init_step=init_step):
            alpha = 20 * np.sqrt(scales[i])
            iter_enabled = i in iter_lst
            thres = float(thres_lst[list(iter_lst).index(i)]) if iter_enabled else 0.0
            x = self.attend_and_excite(x, model, s_in * sig[i], cond, batch, alpha, iter_enabled, thres)
            den = torch.empty_like(x)
            stepper.step(x, sig[i], sig[i + 1], emit_maps=True, denoised=den)
            ll = model.loss_fn.get_min_local_loss(stepper.unet.attn_map_cache, batch["mask"], batch["seg_mask"], cond_only=True)
            local_losses.append(float(ll.mean()))
            if detailed and i == mid:
                attn_map = stepper.unet.save_attn_map(save_name=name, tokens=batch["label"][0])
                self.save_segment_map(attn_map, tokens=batch["label"][0], save_name=name)
            inter = torch.clamp((model.decode_first_stage(den) + 1.0) / 2.0, min=0.0, max=1.0)[0]
            inters.append((inter.float().cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8))
        stepper.check()
        print(f"Local losses: {local_losses}")
        self.last_local_losses, self.last_inters = local_losses, inters
        self.last_aae_stats = f"{self.aae_evaluations - evals0} gradient evaluations"
        try:
            import imageio
            os.makedirs("./temp/inters", exist_ok=True)
            imageio.mimsave(f"./temp/inters/{name}.gif", inters, "GIF", duration=0.02)
        except ImportError:                                            # (no imageio in this image: the frames stay on the sampler)
            pass
        return x

    # ------------------------------------------------------------------------------- batches in flight
    def sample_in_flight(self, model, xs, conds, ucs, init_step=0, deferred_checks: Optional[list] = None,
                         streams: Optional[list] = None):
        """run the sampling loops of SEVERAL independent batches concurrently (one launch stream + one set of
        hipGraphs each, every stream planned for its share of the CUs) and return their latents.

        One launch stream cannot keep the chip busy through every kernel's ramp-up, epilogue and tail; measured on
        MI355X (512x512, batch 4): 14.0 ms per sampler step for one batch at a time, 10.9 ms per step and batch with
        two batches in flight, 10.1 with three (pipeline.IN_FLIGHT).  Falls back to one batch after the other when
        graphs are unavailable.

        ``deferred_checks``: a list that receives the runners' error-word checks instead of running them here (each
        check synchronises its stream) — a caller that keeps enqueuing work behind this call (pipeline.predict_many)
        runs them once at its own synchronisation point.
        ``streams``: launch streams to replay on, one per batch (default: the runners' capture streams).  A caller
        that already owns one stream per batch passes them so that no further streams are active: hipStreams share a
        small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two busy streams on one queue serialise."""
        n = len(xs)
        if n == 1 or not self.use_graphs:
            return [self(model, x, cond=c, uc=u, init_step=init_step) for x, c, u in zip(xs, conds, ucs)]
        self._check_fast_path()
        sig = self._host_sigmas(None)
        steps = list(self.get_sigma_gen(len(sig), init_step=init_step))
        cache = self.__dict__.setdefault("_in_flight", {})
        fp = weights_fingerprint(model)
        runners = []
        for slot, (x, c, u) in enumerate(zip(xs, conds, ucs)):
            require_gpu(x, "EulerEDMSampler")
            u = default(u, c)
            key = (slot, n, id(model), tuple(x.shape), len(sig), float(self.guider.scale), tuple(sig[:2]), x.device.index)
            gs = cache.get(key)
            if gs is not None and gs.fingerprint != fp:        # weights changed: the captured pointers are stale
                cache.pop(key)
                gs = None
            if gs is None or not gs.rebind(c, u):
                gs = _GraphedSteps(model, c, u, x.shape[0], x.shape[2:], self.guider.scale, sig, cu_share=n)
                cache.pop(key, None)
                while len(cache) >= 6:                      # every runner owns a memory pool: keep the newest few
                    cache.pop(next(iter(cache)))
                cache[key] = gs
            gs.x.copy_(x.float())
            gs.x.mul_((1.0 + sig[0] ** 2.0) ** 0.5)
            for i in steps:
                if i not in gs.graphs:
                    gs._capture(i)
            runners.append(gs)
        main = torch.cuda.current_stream()
        lanes = list(streams[:n]) if streams is not None and len(streams) >= n else [gs.capture_stream for gs in runners]
        for lane in lanes:
            lane.wait_stream(main)
        for i in steps:                                  # interleaved launches: every queue stays fed
            for gs, lane in zip(runners, lanes):
                with torch.cuda.stream(lane):
                    gs.graphs[i].replay()
        for lane in lanes:
            main.wait_stream(lane)
        outs = [gs.x.clone() for gs in runners]
        for gs in runners:
            if deferred_checks is not None:
                deferred_checks.append(gs.st.check)
            else:
                gs.st.check()
        return outs

    def sample_lane(self, model, x, cond, uc, slot: int, n_lanes: int, init_step=0, deferred_checks: Optional[list] = None):
        """the sampling loop of ONE batch on the CURRENT stream, as lane ``slot`` of ``n_lanes`` free-running lanes
        (pipeline.predict_many): rebind this lane's runner to the batch, enqueue all graph replays, return the latent — no
        host synchronisation and no event shared with the other lanes, so a lane runs condition -> sample -> decode back to
        back while the launch thread is already feeding the next lane.  Launches are planned for 1 / n_lanes of the CUs
        (cu_share), as in ``sample_in_flight``.  The runner's error-word check goes to ``deferred_checks``."""
        if n_lanes <= 1 or not self.use_graphs:
            return self(model, x, cond=cond, uc=uc, init_step=init_step)
        self._check_fast_path()
        require_gpu(x, "EulerEDMSampler")
        uc = default(uc, cond)
        sig = self._host_sigmas(None)
        steps = list(self.get_sigma_gen(len(sig), init_step=init_step))
        cache = self.__dict__.setdefault("_in_flight", {})
        fp = weights_fingerprint(model)
        key = (slot, n_lanes, id(model), tuple(x.shape), len(sig), float(self.guider.scale), tuple(sig[:2]), x.device.index)
        # a runner that leaves the cache may still have graph replays queued on a lane stream (nothing here synchronises the host):
        # it is parked in ``_retired`` — its graphs and private memory pool stay alive — until the caller has synchronised
        # (pipeline.predict_many -> release_retired); runners of the current call's lanes are never evicted
        # Bounded without the caller's help too (round 6): every runner records an event behind its last replay; parked runners whose
        # event has completed are dropped on the next call — a caller that uses sample_lane directly, or an exception between eviction
        # and release, no longer keeps retired hipGraphs and their memory pools alive indefinitely.
        retired = self.__dict__.setdefault("_retired", [])
        retired[:] = [r for r in retired if getattr(r, "last_replay", None) is not None and not r.last_replay.query()]
        gs = cache.get(key)
        if gs is not None and gs.fingerprint != fp:            # weights changed: the captured pointers are stale
            retired.append(cache.pop(key))
            gs = None
        if gs is None or not gs.rebind(cond, uc):
            gs = _GraphedSteps(model, cond, uc, x.shape[0], x.shape[2:], self.guider.scale, sig, cu_share=n_lanes)
            if key in cache:
                retired.append(cache.pop(key))
            victims = [k for k in cache if not (k[1] == n_lanes and k[2] == id(model) and k[3] == tuple(x.shape))]
            while len(cache) >= 8 and victims:                  # every runner owns a memory pool: keep the newest few
                retired.append(cache.pop(victims.pop(0)))
            cache[key] = gs
        lane = torch.cuda.current_stream()
        missing = [i for i in steps if i not in gs.graphs]
        if missing:
            # first use of this lane / shape: capture on the runner's own stream (a capture synchronises the device once)
            gs.capture_stream.wait_stream(lane)
            for i in missing:
                gs._capture(i)
            lane.wait_stream(gs.capture_stream)
        gs.x.copy_(x.float())
        gs.x.mul_((1.0 + sig[0] ** 2.0) ** 0.5)
        for i in steps:
            gs.graphs[i].replay()
        out = gs.x.clone()
        gs.last_replay = torch.cuda.Event()
        gs.last_replay.record(lane)
        if deferred_checks is not None:
            if gs.st.check not in deferred_checks:
                deferred_checks.append(gs.st.check)
        else:
            gs.st.check()
        return out

    def release_retired(self) -> None:
        """drop the runners sample_lane took out of its cache — call after the lanes' streams have been synchronised"""
        self.__dict__.get("_retired", []).clear()

    def _run_graphed(self, model, x, cond, uc, sig, init_step):
        """replay (capturing on first use) the hipGraphs of this sampling configuration; None -> eager launches"""
        key = (id(model), tuple(x.shape), len(sig), float(self.guider.scale), tuple(sig[:2]), x.device.index)
        cache = self.__dict__.setdefault("_graphed", {})
        try:
            gs = cache.get(key)
            if gs is not None and gs.fingerprint != weights_fingerprint(model):
                gs = None                                       # weights changed under the captured graphs
            if gs is None or not gs.rebind(cond, uc):
                cache.clear()                                   # one configuration at a time (each holds a memory pool)
                gs = _GraphedSteps(model, cond, uc, x.shape[0], x.shape[2:], self.guider.scale, sig)
                cache[key] = gs
            return gs.run(x, self.get_sigma_gen(len(sig), init_step=init_step))
        except RuntimeError as e:
            if not _is_capture_failure(e):                      # kernel status errors, OOM, ...: never hidden
                raise
            if not self.__dict__.get("_graph_warned"):
                print(f"[udifftext_amd] hipGraph capture unavailable ({e}); using eager launches", file=sys.stderr)
                self._graph_warned = True
            self.use_graphs = False
            cache.clear()
            return None

    def invalidate_graphs(self) -> None:
        """drop every captured hipGraph (explicit hook; the caches also notice changed weights by fingerprint)"""
        self.__dict__.pop("_graphed", None)
        self.__dict__.pop("_in_flight", None)
