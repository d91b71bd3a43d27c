"""cvcl_bf16_to_f32 / cvcl_f32_to_bf16 (include/cvcl_hip.h): the casts either side of the spatial head's 1x1 projection when the trunk
stores bf16 (reference multimodal/multimodal.py:181-185 runs the projection on the trunk's own dtype; here the head is fp32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_casts_round_to_nearest_even_and_round_trip(dev):
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096 * 8, generator=g) * torch.logspace(-20, 20, 4096 * 8, base=2.0)
    x[:8] = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.01171875, float("inf"), -float("inf"), 3.3895313892515355e38])  # ties, inf, near max
    xd = x.to(dev)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_f32_to_bf16(H.ptr(xd, torch.float32), H.ptr(y), x.numel(), H.stream_ptr()), "cvcl_f32_to_bf16")
    assert torch.equal(y.cpu(), x.bfloat16())                                  # torch rounds to nearest even as well
    z = torch.empty(x.shape, dtype=torch.float32, device=dev)
    H.check(H.lib().cvcl_bf16_to_f32(H.ptr(y), H.ptr(z), x.numel(), H.stream_ptr()), "cvcl_bf16_to_f32")
    assert torch.equal(z.cpu(), x.bfloat16().float())
    assert H.lib().cvcl_f32_to_bf16(H.ptr(xd, torch.float32), H.ptr(y), 7, H.stream_ptr()) != 0      # n % 8 != 0 is refused
