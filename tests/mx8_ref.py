"""Test-side torch restatement of the MX8 activation format (udifftext_amd/csrc/common.h "MX8 activations"): OCP e4m3 elements
+ one E8M0 scale per 32 consecutive columns of a row, scales stored as int32 [K / 128, M] (byte j of dword (t, m) = block 4 t + j).
Checker only — the product path never quantises on the host."""
import torch


def scale_bytes(amax: torch.Tensor) -> torch.Tensor:
    """the kernels' rule (mx8_scale_byte): smallest power of two 2^(s - 127) with amax / 2^(s - 127) <= 448"""
    t = (amax.float() * torch.tensor(0.57142866, dtype=torch.float32, device=amax.device))
    e = (t.view(torch.int32) >> 23) & 0xFF
    return torch.clamp(e - 7, min=0)


def encode(x: torch.Tensor):
    """fp32 / bf16 [M, K] (K % 128 == 0) -> (uint8 [M, K], int32 [K / 128, M])"""
    M, K = x.shape
    assert K % 128 == 0
    xb = x.float().reshape(M, K // 32, 32)
    s = scale_bytes(xb.abs().amax(dim=2))                                   # [M, K/32]
    inv = ((254 - s) << 23).to(torch.int32).view(torch.float32)
    q = (xb * inv[:, :, None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(M, K)
    s4 = s.reshape(M, K // 128, 4).to(torch.int32)
    packed = s4[..., 0] | (s4[..., 1] << 8) | (s4[..., 2] << 16) | (s4[..., 3] << 24)
    return q.contiguous(), packed.t().contiguous()


def decode(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(uint8 [M, K], int32 [ceil(K / 128), M]) -> fp32 [M, K]"""
    M, K = q.shape
    kt = scale.shape[0]
    s = torch.stack([(scale >> (8 * j)) & 0xFF for j in range(4)], dim=2)   # [kt, M, 4]
    s = s.permute(1, 0, 2).reshape(M, kt * 4)[:, : (K + 31) // 32]
    mul = (s << 23).to(torch.int32).view(torch.float32)                      # 2^(s - 127)
    mul = torch.where(s == 0, torch.full_like(mul, 2.0 ** -127), mul)
    v = q.view(torch.float8_e4m3fn).float().reshape(M, -1)
    Kp = s.shape[1] * 32
    if Kp != K:
        v = torch.nn.functional.pad(v, (0, Kp - K))
    return (v.reshape(M, -1, 32) * mul[:, :, None]).reshape(M, Kp)[:, :K]
