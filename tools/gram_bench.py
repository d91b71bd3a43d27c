"""Timing of the Gram-statistics route (cvcl_conv1x1_gram + cvcl_bn_from_gram) against the statistics-only GEMM pass + cvcl_bn_finalize
it replaces, on the trunk's shapes at B = 256 (layer-1 / layer-2 conv3, layer1.0 downsample)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB: evicts the caches between variants
for name, M, K, N in (("layer1 conv3", 802816, 128, 256), ("layer2 conv3", 200704, 256, 512), ("layer1.0 downsample", 802816, 64, 256)):
    a = (torch.randn(M, K, device=dev) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    plain = K == 64
    sc, sh = (None, None) if plain else (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.2)
    gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    rm, rv, nbt = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    scale, shift = torch.empty(N, device=dev), torch.empty(N, device=dev)
    nb = H.lib().cvcl_conv1x1_gram_workspace_bytes(K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = C.c_void_p()
    def gram():
        H.check(H.lib().cvcl_conv1x1_gram(H.ptr(a), K, M, K, H.ptr(sc), H.ptr(sh), int(not plain), H.ptr(ws), nb, C.byref(out), H.stream_ptr()), "gram")
        H.check(H.lib().cvcl_bn_from_gram(out, K, M, H.ptr(w), K, N, H.ptr(gamma), H.ptr(beta), H.ptr(rm), H.ptr(rv), H.ptr(nbt), 0.1, 1e-5,
                                          H.ptr(scale), H.ptr(shift), None, 0, None, H.stream_ptr()), "bn_from_gram")
    def gram_only():
        H.check(H.lib().cvcl_conv1x1_gram(H.ptr(a), K, M, K, H.ptr(sc), H.ptr(sh), int(not plain), H.ptr(ws), nb, C.byref(out), H.stream_ptr()), "gram")
    rows = H.gemm_stats_rows(H.BF16, M, N, K, prologue=not plain, a_relu=not plain)
    st = torch.empty(max(rows, 1024), 2, N, device=dev)
    ga = H.GemmArgs()
    ga.A, ga.W, ga.C = H.ptr(a), H.ptr(w), None
    ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldc = M, N, K, K, K, N
    if not plain:
        ga.a_scale, ga.a_shift, ga.a_relu = H.ptr(sc), H.ptr(sh), 1
    ga.stats, ga.stats_rows = H.ptr(st), st.shape[0]
    def stats_pass():
        H.check(H.lib().cvcl_gemm(H.BF16, C.byref(ga), H.stream_ptr()), "stats pass")
        H.check(H.lib().cvcl_bn_finalize(H.ptr(st), rows, M, H.ptr(gamma), H.ptr(beta), H.ptr(rm), H.ptr(rv), H.ptr(nbt), 0.1, 1e-5, H.ptr(scale),
                                         H.ptr(shift), None, N, H.stream_ptr()), "finalize")
    def cold(f):
        def g():
            big.zero_()
            f()
        return g
    z = timeit(cold(lambda: None))
    print(f"{name:22s} M={M} K={K} N={N}: Gram route {timeit(gram):7.1f} us (gram + reduce alone {timeit(gram_only):7.1f}) | statistics pass + finalize "
          f"{timeit(stats_pass):7.1f} us | behind a 1 GiB memset: Gram {timeit(cold(gram)) - z:7.1f}, pass {timeit(cold(stats_pass)) - z:7.1f}")
