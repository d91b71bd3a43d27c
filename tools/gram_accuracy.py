"""Accuracy of BatchNorm (scale, shift) from the Gram route against float64 at the trunk's row counts, on post-ReLU-like operands whose
column means are large against their spread (where sum y^2 / M - mean^2 cancels): relative error of the variance and of scale, next to the
statistics-only GEMM pass + cvcl_bn_finalize with a centre near the mean (the route it replaced)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for name, M, K, N, plain in (("layer1.0 downsample", 802816, 64, 256, True), ("layer1 conv3", 802816, 128, 256, False), ("layer2 conv3", 200704, 256, 512, False)):
    col_mean = (0.5 + torch.rand(K, generator=g)).to(dev)
    col_std = (0.2 + 0.5 * torch.rand(K, generator=g)).to(dev)
    a = torch.relu(torch.randn(M, K, device=dev) * col_std + col_mean).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(dev)
    gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    # float64 truth in row chunks
    s1 = torch.zeros(N, dtype=torch.float64, device=dev); s2 = torch.zeros(N, dtype=torch.float64, device=dev)
    wd = w.double()
    for i in range(0, M, 65536):
        y = a[i:i + 65536].double() @ wd.t()
        s1 += y.sum(0); s2 += (y * y).sum(0)
    mean = s1 / M
    var = torch.zeros(N, dtype=torch.float64, device=dev)
    for i in range(0, M, 65536):
        y = a[i:i + 65536].double() @ wd.t()
        var += ((y - mean) ** 2).sum(0)
    var /= M
    ref_scale = 1.0 / torch.sqrt(var + 1e-5)
    nb = H.lib().cvcl_conv1x1_gram_workspace_bytes(K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = C.c_void_p()
    H.check(H.lib().cvcl_conv1x1_gram(H.ptr(a), K, M, K, None, None, 0, H.ptr(ws), nb, C.byref(out), H.stream_ptr()), "gram")
    scale, shift = torch.empty(N, device=dev), torch.empty(N, device=dev)
    mom = torch.zeros(2, 2048, device=dev)
    centre = mean.float()
    H.check(H.lib().cvcl_bn_from_gram(out, K, M, H.ptr(w), K, N, H.ptr(gamma), H.ptr(beta), None, None, None, 0.1, 1e-5, H.ptr(scale), H.ptr(shift),
                                      H.ptr(mom), 2048, H.ptr(centre), H.stream_ptr()), "from_gram")
    torch.cuda.synchronize()
    e_gram = ((scale.double() - ref_scale).abs() / ref_scale).max().item()
    canc = ((s2 / M) / var).max().item()
    # the old route: statistics of the rounded, centred product
    rows = H.gemm_stats_rows(H.BF16, M, N, K, prologue=False, a_relu=False)
    st = torch.empty(max(rows, 1024), 2, N, device=dev)
    ga = H.GemmArgs()
    ga.A, ga.W, ga.C = H.ptr(a), H.ptr(w), None
    ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldc = M, N, K, K, K, N
    ga.stats, ga.stats_rows = H.ptr(st), st.shape[0]
    ga.centre = H.ptr(centre)
    s2_, h2_ = torch.empty(N, device=dev), torch.empty(N, device=dev)
    H.check(H.lib().cvcl_gemm(H.BF16, C.byref(ga), H.stream_ptr()), "stats pass")
    H.check(H.lib().cvcl_bn_finalize(H.ptr(st), rows, M, H.ptr(gamma), H.ptr(beta), None, None, None, 0.1, 1e-5, H.ptr(s2_), H.ptr(h2_), H.ptr(centre), N,
                                     H.stream_ptr()), "finalize")
    torch.cuda.synchronize()
    e_pass = ((s2_.double() - ref_scale).abs() / ref_scale).max().item()
    e_shift_g = (shift.double() - (0 - (mean - centre.double()) * ref_scale)).abs().max().item()
    e_shift_p = (h2_.double() - (0 - (mean - centre.double()) * ref_scale)).abs().max().item()
    print(f"{name}: E[y^2]/var up to {canc:.1f}; scale rel err  Gram {e_gram:.2e}  pass {e_pass:.2e};  shift abs err  Gram {e_shift_g:.2e}  pass {e_shift_p:.2e}")
