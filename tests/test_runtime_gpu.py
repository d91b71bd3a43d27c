"""GPU: libcvcl_hip shares torch's HIP runtime (one libamdhip64 mapped) and runs on torch's stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_hip_runtime_and_native_lib_loaded(dev):
    from multimodal import _hip
    _hip.load()
    torch.zeros(1, device=dev)
    maps = open("/proc/self/maps").read()
    hip = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
    cv = sorted({l.split()[-1] for l in maps.splitlines() if "libcvcl_hip" in l})
    print("HIP runtimes mapped:", hip)
    print("native lib:", cv)
    assert len(hip) == 1, hip
    assert len(cv) == 1 and cv[0].endswith("multimodal-baby_amd/lib/libcvcl_hip.so")


def test_runs_on_side_stream(dev):
    from multimodal import ops
    x = torch.randn(64, 128, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = ops.l2_normalize(x)
    s.synchronize()
    assert torch.allclose(y.norm(dim=1), torch.ones(64, device=dev), atol=1e-5)
