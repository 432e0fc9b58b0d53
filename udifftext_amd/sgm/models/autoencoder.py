"""AutoencoderKL on the gfx950 kernels, reference interface (sgm/models/autoencoder.py:17-72,282-321):
``encode(x)`` takes an NCHW fp32 image in [-1, 1] and returns the posterior (``AutoencoderKL``) or a posterior
sample (``AutoencoderKLInferenceWrapper``); ``decode(z)`` returns an NCHW fp32 image.  Training-side methods
(GAN loss, optimisers, EMA, logging) are out of scope.
"""
from __future__ import annotations

import re
from typing import Any

import torch
import torch.nn as nn

from udifftext_amd import ops, packing

from ..modules import hipnn as H
from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution
from ..util import instantiate_from_config, require_gpu


def _load_checkpoint(path: str) -> dict:
    if path.endswith("ckpt"):
        return torch.load(path, map_location="cpu")["state_dict"]
    if path.endswith("safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    raise NotImplementedError(path)


class AbstractAutoencoder(nn.Module):
    def __init__(self, ema_decay=None, monitor=None, input_key="jpg", ckpt_path=None, ignore_keys=()):
        super().__init__()
        if ema_decay is not None:
            raise NotImplementedError("EMA is a training feature (out of scope)")
        self.input_key = input_key
        self.use_ema = False
        if monitor is not None:
            self.monitor = monitor

    def init_from_ckpt(self, path: str, ignore_keys=tuple()) -> None:
        sd = _load_checkpoint(path)
        for k in list(sd.keys()):
            if any(re.match(ik, k) for ik in ignore_keys):
                print(f"Deleting key {k} from state_dict.")
                del sd[k]
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if missing:
            print(f"Missing Keys: {missing}")
        if unexpected:
            print(f"Unexpected Keys: {unexpected}")

    def get_input(self, batch) -> Any:
        return batch[self.input_key]

    def encode(self, *args, **kwargs):
        raise NotImplementedError

    def decode(self, *args, **kwargs):
        raise NotImplementedError


class AutoencodingEngine(AbstractAutoencoder):
    def __init__(self, *args, encoder_config, decoder_config, loss_config, regularizer_config, optimizer_config=None,
                 lr_g_factor: float = 1.0, **kwargs):
        ckpt_path = kwargs.pop("ckpt_path", None)
        super().__init__(*args, **kwargs)
        self.encoder = instantiate_from_config(encoder_config)
        self.decoder = instantiate_from_config(decoder_config)
        self.loss = instantiate_from_config(loss_config)
        self.regularization = instantiate_from_config(regularizer_config)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)


class AutoencoderKL(AutoencodingEngine):
    def __init__(self, embed_dim: int, **kwargs):
        ddconfig = kwargs.pop("ddconfig")
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", ())
        super().__init__(encoder_config={"target": "torch.nn.Identity"}, decoder_config={"target": "torch.nn.Identity"},
                         regularizer_config={"target": "torch.nn.Identity"}, loss_config=kwargs.pop("lossconfig"),
                         **kwargs)
        assert ddconfig["double_z"]
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        zc = ddconfig["z_channels"]
        self.quant_conv = H.Conv2d(2 * zc, 2 * embed_dim, 1)
        self.post_quant_conv = H.Conv2d(embed_dim, zc, 1)
        self.embed_dim = embed_dim
        # 1x1 convs consume / produce 64-channel-padded activations (K of the GEMM is a multiple of 64)
        self.encoder.conv_out.n_pad = packing.KPAD
        self.post_quant_conv.n_pad = packing.KPAD
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """NCHW fp32 image -> fp32 NHWC moments [B, h, w, 8]"""
        require_gpu(x, "AutoencoderKL.encode")
        assert not self.training, f"{self.__class__.__name__} only supports inference currently"
        xin = ops.nchw_to_nhwc(x.float().contiguous(), packing.KPAD)
        h = self.encoder(xin)
        return self.quant_conv(h, flags=H.GEMM_OUT_F32)

    def encode(self, x):
        return DiagonalGaussianDistribution(self.encode_moments(x))

    def decode(self, z, **decoder_kwargs):
        require_gpu(z, "AutoencoderKL.decode")
        zin = ops.nchw_to_nhwc(z.float().contiguous(), packing.KPAD)
        dec = self.decoder(self.post_quant_conv(zin))
        return ops.nhwc_to_nchw(dec, self.decoder.out_ch)


class AutoencoderKLInferenceWrapper(AutoencoderKL):
    def encode(self, x):
        return super().encode(x).sample()
