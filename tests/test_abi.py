"""CPU: libcvcl_hip.so builds, loads and exports every symbol include/cvcl_hip.h declares; argument
validation answers without touching a GPU; the product path refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "cvcl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cvcl_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("cvcl_build", os.path.join(ROOT, "multimodal-baby_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(verbose=False)
    from multimodal import _hip
    return _hip


def test_every_declared_symbol_is_bound_and_exported(lib):
    declared = _declared()
    assert declared, "header parse failed"
    assert sorted(lib.SIGNATURES.keys()) == declared
    l = lib.load()
    for name in declared:
        assert hasattr(l, name), name
    assert l.cvcl_abi_version() == lib.ABI_VERSION


def test_argument_validation_without_gpu(lib):
    l = lib.load()
    a = lib.GemmArgs()
    assert l.cvcl_gemm(lib.F32, a, None) == -1            # CVCL_EINVAL: null operands
    assert b"null" in l.cvcl_last_error()
    assert l.cvcl_l2norm_fwd(None, None, None, 0, 0, 1e-12, None) == -1
    assert l.cvcl_infonce_workspace_bytes(256) == 6 * 256 * 4


def test_no_cpu_fallback(lib):
    from multimodal import ops
    with pytest.raises(lib.CvclError):
        ops.l2_normalize(torch.randn(4, 8))
    with pytest.raises(lib.CvclError):
        lib.gemm(torch.randn(4, 8), torch.randn(4, 8))
