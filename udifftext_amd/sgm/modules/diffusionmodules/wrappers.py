"""Network wrapper: concatenates the ``concat`` conditioning on the channel axis and routes the
cross-attention context (reference sgm/modules/diffusionmodules/wrappers.py:8-35)."""
import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        # compile_model is accepted for config compatibility; kernels are hand-written HIP, nothing to trace
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        if "concat" in c:
            x = torch.cat((x, c["concat"]), dim=1)
        return self.diffusion_model(x, timesteps=t, t_context=c.get("t_crossattn", None),
                                    v_context=c.get("v_crossattn", None), y=c.get("vector", None), **kwargs)
