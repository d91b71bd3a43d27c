"""CVCL contrastive train-step benchmark on MI355X (driver contract: see the task prompt / DESIGN.md section 7).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts that torch.distributed.run command itself as a
child process (before this process has touched the GPU) and exits with its code.

A step = one pass of the hot path over one batch of synthetic input already resident in HBM: image trunk forward,
text encoder, L2 normalise, [RCCL feature all-gather], similarity logits, symmetric InfoNCE, backward of the trainable
set, [RCCL gradient all-reduce], AdamW.  Weak scaling (the per-GPU batch is fixed).

    c2 (default)  BASELINE.json configs[1]: frozen ResNeXt-50 32x4d (BN train mode) + embedding text encoder, bf16, 256 / GPU
    c4            configs[3]: frozen DINO ViT-B/16 (bf16 MFMA linears) + trainable transformer text encoder, 256 / GPU
    c5            configs[4]: c4 with e4m3 weights / activations in the ViT linears (scaled MFMA); --batch grows the per-GPU batch

Rank 0 prints ONE JSON line with the throughput, the measured deviation of the benchmarked precision from the exact-fp32
parity mode on the same batch (``logits_rel_vs_fp32``), the roofline of the dominant kernel (timed live with HIP events on
the launch stream in a second pass of the same steps, so the events do not perturb the headline number) and a CPU
baseline (the oracle restatement of the reference step on the host cores, bounded sample).
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

METRIC = "image-text pairs/sec, CVCL ResNeXt+embed 224², bs256, 1/2/4/8 MI355X"
PER_GPU_BATCH = 256
EMBEDDING_DIM = 512
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled MFMA)
PROFILE_ROUND = "r06"

WORKLOADS = {
    "c2": "C2 = BASELINE configs[1]: CVCL saycam_contrastive, frozen random-init ResNeXt-50 32x4d (BN train mode) + embedding "
          "mean-pool text encoder, E=512, L2-normalised, fixed tau 0.07, 224x224 frames + 3-word utterances; full step = fwd + "
          "InfoNCE + bwd(fc, embedding) + AdamW",
    "c4": "C4 = BASELINE configs[3]: saycam_contrastive_transformer, frozen random-init DINO ViT-B/16 (bf16 MFMA linears) + "
          "trainable one-layer transformer text encoder (learned positions, dropout 0.1), E=512, L2-normalised, fixed tau 0.07; "
          "full step = fwd + InfoNCE + bwd(head, text encoder) + AdamW",
    "c4p14": "C4 at the reference's own ViT (multimodal/multimodal.py:135 hard-codes dino_sfp_vitb14): frozen random-init DINO ViT-B/14 "
             "(257 tokens, 46.3 GFLOP per pair; bf16 MFMA linears) + trainable transformer text encoder, otherwise as C4",
    "c5": "C5 = BASELINE configs[4]: C4 with e4m3 weights (per-channel scales) and e4m3 activations (per-token / MX block "
          "scales) in the four linears of every ViT block on v_mfma_scale_f32_32x32x64_f8f6f4; bf16 residual stream",
}


def c2_args():
    return argparse.Namespace(
        embedding_type="flat", embedding_dim=EMBEDDING_DIM, pretrained_cnn=False, cnn_model="resnext50_32x4d",
        cnn_dino=False, vit_dino=False, finetune_cnn=False, text_encoder="embedding", captioning=False,
        attention=False, attention_gate=False, crange=1, dropout_i=0.5, dropout_o=0.0, pos_embed_type="no_pos_embed",
        normalize_features=True, sim="max", temperature=0.07, fix_temperature=True, tie=True, bias=True,
        optimizer=torch.optim.AdamW, lr=1e-4, weight_decay=0.1, lr_scheduler=False, lambda_mm=1.0, lambda_lm=0.0,
        lambda_ar=0.0, optimize_unused=True, local_negatives=False)


def c4_args():
    a = c2_args()
    a.vit_dino, a.text_encoder, a.pos_embed_type, a.dropout_i = True, "transformer", "learned", 0.0
    return a


def patch_of(config):
    return 14 if config.endswith("p14") else 16


def build_model(config, device, precision=None, seed=0, patch=None):
    """The benchmarked module with random-init weights: -> (lit, vision_encoder, optimizer).  ``precision`` None = the
    configuration's own (bf16; fp8 linears for c5)."""
    import multimodal.multimodal as mm
    from multimodal import vision_transformer_dino_mugs as vits
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel
    torch.manual_seed(seed)
    args = c2_args() if config == "c2" else c4_args()
    patch = patch or patch_of(config)
    orig = mm.load_model
    if config != "c2":      # BASELINE names ViT-B/16; the reference hard-codes vitb14 (multimodal.py:135), vit_base(16) exists (:287)
        mm.load_model = lambda name, pretrained: vits.vit_base(patch_size=patch, num_classes=0)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ve = VisionEncoder(args)
            te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
            lit = MultiModalLitModel(ve, te, args)
    finally:
        mm.load_model = orig
    lit.to(device)
    lit.set_precision(precision or ("fp8" if config == "c5" else "bf16"))
    lit.train()                                           # Lightning keeps .train(): BN uses batch statistics
    return lit, ve, lit.configure_optimizers()


@contextlib.contextmanager
def frozen_trunk_cached(ve, cfg, images):
    """The frozen image trunk replaced by its own (cached) output for ``images``: what is left of a step is the TRAINABLE TAIL --
    fc / head, text encoder forward + backward, L2 normalise, logits, InfoNCE forward + backward, AdamW -- with nothing of the
    trunk on the GPU beside it (``tail_ms_per_step``; the trunk stream must be off)."""
    from multimodal import vit_hip
    with torch.no_grad():
        if cfg == "c2":
            pooled, fmap = ve.model.trunk(images)
            pooled, fmap = pooled.clone(), fmap.clone()
        else:
            cls = vit_hip._vit_forward(ve.model, images, None).clone()
    torch.cuda.synchronize()
    if cfg == "c2":
        ve.model.trunk = lambda x, defer_wait=False: (pooled, fmap)
    else:
        orig = vit_hip._vit_forward
        vit_hip._vit_forward = lambda model, x, slot: cls
    try:
        yield
    finally:
        if cfg == "c2":
            del ve.model.trunk
        else:
            vit_hip._vit_forward = orig


def synthetic_batch_on_device(batch, seed, device, vocab=2350):
    """rand -> ImageNet normalise; <sos> w1 w2 w3 <eos>, generated once on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    img = torch.rand(batch, 3, 224, 224, generator=g, device=device)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
    img = ((img - mean) / std).contiguous()
    words = torch.randint(4, vocab, (batch, 3), generator=g, device=device)
    tok = torch.cat([torch.full((batch, 1), 2, device=device), words, torch.full((batch, 1), 3, device=device)], 1).long()
    ln = torch.full((batch,), 5, dtype=torch.long, device=device)
    return img, tok.contiguous(), ln


def structured_batch_on_device(batch, seed, device, vocab=2350):
    """A second synthetic input: every frame is a smooth random field -- 3 x 7 x 7 uniform noise upsampled bicubically to
    224 x 224 -- with its own contrast and brightness, clamped to [0, 1], then ImageNet-normalised: the samples differ at every
    scale (iid pixel noise makes them nearly identical after the stem).  Used by structured_parity."""
    g = torch.Generator(device=device).manual_seed(seed)
    coarse = torch.rand(batch, 3, 7, 7, generator=g, device=device)
    img = torch.nn.functional.interpolate(coarse, size=(224, 224), mode="bicubic", align_corners=False)
    contrast = 0.3 + 1.4 * torch.rand(batch, 1, 1, 1, generator=g, device=device)
    bright = 0.25 + 0.5 * torch.rand(batch, 1, 1, 1, generator=g, device=device)
    img = ((img - 0.5) * contrast + bright).clamp_(0.0, 1.0)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
    img = ((img - mean) / std).contiguous()
    words = torch.randint(4, vocab, (batch, 3), generator=g, device=device)
    tok = torch.cat([torch.full((batch, 1), 2, device=device), words, torch.full((batch, 1), 3, device=device)], 1).long()
    ln = torch.full((batch,), 5, dtype=torch.long, device=device)
    return img, tok.contiguous(), ln


def structured_parity(lit, ve, batch_size, device, rank=0, yardstick=True, gamma3=0.25):
    """A WELL-CONDITIONED parity point for the ResNeXt configuration: bf16 vs fp32 (and torch autocast vs torch fp32) with the
    residual-branch gains set to ``gamma3`` (every Bottleneck's bn3.weight: random init leaves them at 1, where each of the 16 blocks
    doubles the signal's variance and the 50-layer train-mode-BatchNorm network amplifies ANY rounding ~100x whatever the input --
    measured: 0.18 on noise frames, 0.16 on smooth frames; trained / zero-init-residual networks sit near 0.1-0.3) on structured
    frames.  There torch's own autocast deviates ~2e-2 and a systematic error of a few per cent in any layer would show.
    Storage centres calibrated on ANOTHER structured batch; weights, BatchNorm buffers and centres restored afterwards."""
    keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
    bn3 = [m.bn3 for m in ve.model.modules() if hasattr(m, "bn3")]
    g_keep = [b.weight.detach().clone() for b in bn3]
    evalb = structured_batch_on_device(batch_size, seed=4242 + rank, device=device)
    calib = structured_batch_on_device(batch_size, seed=1717 + rank, device=device)
    gn, lit.model.global_negatives = lit.model.global_negatives, False
    try:
        with torch.no_grad():
            for b in bn3:
                b.weight.fill_(gamma3)
        if hasattr(ve.model, "recalibrate_centres"):
            ve.model.recalibrate_centres()
        with torch.no_grad():
            lit.model(calib[0], calib[1], calib[2])
        lit.load_state_dict(keep, strict=False)
        out = {k: float(f"{v:.4g}") for k, v in logits_vs_fp32(lit, evalb, "bf16").items()}
        if yardstick:
            ty, _ = torch_yardstick(lit, evalb)
            out["torch_autocast_bf16_vs_torch_fp32"] = {k: float(f"{v:.4g}") for k, v in ty.items()}
    finally:
        with torch.no_grad():
            for b, g in zip(bn3, g_keep):
                b.weight.copy_(g)
        lit.model.global_negatives = gn
        lit.load_state_dict(keep, strict=False)
        if hasattr(ve.model, "recalibrate_centres"):
            ve.model.recalibrate_centres()
    out["setting"] = (f"every Bottleneck's bn3.weight = {gamma3} (residual-branch gain of a trained / zero-init-residual network; random init "
                      "leaves 1.0), frames = smooth random fields (3 x 7 x 7 noise, bicubic to 224 x 224) with per-frame contrast / brightness")
    return out


def logits_vs_fp32(lit, batch, precision):
    """Deviation of the benchmarked precision from the exact-fp32 parity mode (the mode that meets the 1e-3 gate against the
    reference forward, multimodal.py:746-794) on the SAME weights and batch, module in train mode as benchmarked (BatchNorm on
    batch statistics): the rank-local logits matrix and InfoNCE loss.  The BatchNorm buffers are restored afterwards."""
    from multimodal import ops
    x, y, ln = batch[0], batch[1], batch[2]
    model = lit.model
    keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
    was_training, gn = lit.training, model.global_negatives
    model.global_negatives = False                          # rank-local B x B logits (no collective in this check)
    te = model.text_embed
    out = {}
    try:
        lit.train()
        te.eval()                                           # dropout off (it would draw different masks in the two runs)
        res = {}
        with torch.no_grad():
            for p in ("32", precision):
                lit.set_precision(p)
                li, _lt = model(x, y, ln)
                loss, _m = ops.infonce(li)
                res[p] = (li.float().clone(), float(loss))
        a, b = res[precision][0].double(), res["32"][0].double()
        out = {"logits_rel_vs_fp32": float((a - b).abs().max() / b.abs().max()),
               "logits_cosine_vs_fp32": float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)),
               "loss_abs_vs_fp32": abs(res[precision][1] - res["32"][1]), "loss_fp32": res["32"][1]}
    finally:
        model.global_negatives = gn
        lit.set_precision(precision)
        lit.train(was_training)
        lit.load_state_dict(keep, strict=False)
    return out


def resnext_gemm_work(B):
    """Algorithmic bytes / flops of the bf16 conv GEMM launches of one ResNeXt-50 forward at batch B (train mode), per
    kernel: every operand element moved once, 2 bytes each; 2*M*N*K flops per launch.  Launch list = what cvcl_resnext50_fwd
    enqueues: conv1 and downsample (A + W + C); conv3 of layers 3-4 (A + W + C); conv3 of layers 1-2 as the Gram launch that
    stands in for its BN3 statistics (A only; M K (K + 32) flops) and the fused BN3 + identity + ReLU pass (A + W + residual + C).  (The two leading stages use the fused tail.)  Which kernel runs a launch mirrors
    the dispatcher of csrc/gemm.hip: gemm_pro = conv3 with the BN2+ReLU operand prologue (layers 1-2), gemm8w = plain operands,
    K >= 256, N % 256 == 0, >= 96 tiles of 256 x 256, N K >= 170 (N + K) (strided-gather downsamples included); the rest on gemm_glds ("gemm").
    -> {kernel: [bytes, flops, launches]} and the totals."""
    fused_stages, pro_stages = 2, 2                  # (lab switches of the library, fixed in the product build: csrc/resnext.hip)
    # layer1.0: the downsample launch is a Gram launch over the block input (statistics only) and the tail pass recomputes the branch
    # from the block input (reads X [M, 64] + W2 instead of the stored [M, 256] branch) -- csrc/resnext.hip
    ds_recompute = os.environ.get("CVCL_GEMM_PRO", "1") != "0"
    per = {"gemm": [0, 0, 0], "gemm8w": [0, 0, 0], "gemm_pro": [0, 0, 0]}

    def add(kernel, nbytes, flops):
        per[kernel][0] += nbytes; per[kernel][1] += flops; per[kernel][2] += 1

    def plain_kernel(m, n, k, gather):
        ok = (k >= 256 and k % 128 == 0 and n % 256 == 0 and -(-m // 256) * (n // 256) >= 96 and n * k >= 170 * (n + k)
              and os.environ.get("CVCL_GEMM8W", "1") != "0")
        return "gemm8w" if ok else "gemm"
    inplanes, h = 64, 56
    for stage, blocks in enumerate((3, 4, 6, 3)):
        planes = 64 << stage
        width, outc = planes * 2, planes * 4
        for bi in range(blocks):
            stride = 2 if (stage > 0 and bi == 0) else 1
            ho = h // stride
            m_in, m_out = B * h * h, B * ho * ho
            m, n, k = m_in, width, inplanes                          # conv1
            add(plain_kernel(m, n, k, False), 2 * (m * k + n * k + m * n), 2 * m * n * k)
            recompute = ds_recompute and stage == 0 and bi == 0
            gram = lambda m, k: (2 * m * k, m * k * (k + 32))        # csrc/bn_gram.hip: one read of the operand, upper-triangle tiles of A^T A
            if bi == 0:                                              # downsample
                m, n, k = m_out, outc, inplanes
                if recompute:                                        # only its BN statistics are needed: Gram launch (timed with the gemm_pro class)
                    add("gemm_pro", *gram(m, k))
                else:
                    add(plain_kernel(m, n, k, stride > 1), 2 * (m * k + n * k + m * n), 2 * m * n * k)
            m, n, k = m_out, outc, width                             # conv3
            pro = stage < pro_stages and width in (128, 256)
            kern = "gemm_pro" if pro else plain_kernel(m, n, k, False)
            if stage < fused_stages:                                 # BN3 statistics (Gram launch; round 2-3: a statistics-only GEMM pass) + fused tail pass
                if pro:
                    add("gemm_pro", *gram(m, k))
                else:
                    add("gemm", 2 * (m * k + n * k), 2 * m * n * k)
                if recompute:                                        # + X and W2 read, + the K2 = 64 product; no residual read
                    add(kern, 2 * (m * k + n * k + m * n + m * inplanes + n * inplanes), 2 * m * n * (k + inplanes))
                else:
                    add(kern if pro else "gemm", 2 * (m * k + n * k + 2 * m * n), 2 * m * n * k)
            else:
                add(kern, 2 * (m * k + n * k + m * n), 2 * m * n * k)
            h, inplanes = ho, outc
    tot = [sum(v[i] for v in per.values()) for i in range(3)]
    return per, tot[0], tot[1], tot[2]


def vit_gemm_work(B, patch=16, D=768, depth=12, mlp=3072, operand_bytes=2):
    """Algorithmic flops / bytes of the ViT-B GEMM launches of one forward at batch B: the patch embedding and the four
    linears of every block (qkv, proj, fc1, fc2).  Bytes: A + W at the operand width, C in bf16."""
    T = (224 // patch) ** 2 + 1
    M = B * T
    shapes = [(B * (T - 1), D, (3 * patch * patch + 7) // 8 * 8)]
    for _ in range(depth):
        shapes += [(M, 3 * D, D), (M, D, D), (M, mlp, D), (M, D, mlp)]
    flops = sum(2 * m * n * k for m, n, k in shapes)
    nbytes = sum(operand_bytes * (m * k + n * k) + 2 * m * n for m, n, k in shapes)
    return nbytes, flops, len(shapes)


def cpu_baseline():
    """The oracle (CPU restatement of the reference step: fwd + InfoNCE + bwd(trainable) + AdamW, BatchNorm through
    F.batch_norm as nn.BatchNorm2d runs it) timed on the host cores as SURVEY.md 8(d) prescribes: C1 (B = 8, E = 128,
    learned temperature, no normalise) for 5 steps and C2 (B = 256, E = 512) for 2 steps, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cvcl_oracle as O
    threads = torch.get_num_threads()

    def run(batch, steps, embed, normalize, warm):
        p = O.cvcl_random_params(embed, seed=0)
        step = O.CpuTrainStep(p, lr=1e-4, weight_decay=0.1, normalize_features=normalize, bn_impl="torch",
                              learn_temperature=not normalize)         # C1 (run.sh:12) learns the temperature, C2 fixes it
        img, tok, ln = O.synthetic_batch(batch, seed=0)
        for _ in range(warm):
            step.step(img, tok, ln)
        t0 = time.perf_counter()
        for _ in range(steps):
            step.step(img, tok, ln)
        return time.perf_counter() - t0

    t1 = run(8, 5, 128, False, 1)                             # also warms the thread pool / allocator for the C2 sample
    t2 = run(PER_GPU_BATCH, 2, EMBEDDING_DIM, True, 0)
    host = f"fp32, torch {torch.__version__} CPU, {threads} threads of {os.cpu_count()} cpus"
    return {"value": round(PER_GPU_BATCH * 2 / t2, 2), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"C2: 2 train steps at batch {PER_GPU_BATCH} (ResNeXt-50 fwd with train-mode BN + embedding + InfoNCE + "
                      f"bwd + AdamW), {host}, {t2:.1f}s",
            "c1": {"value": round(8 * 5 / t1, 2), "unit": "pairs/s", "cores": threads,
                   "sample": f"C1 = BASELINE configs[0]: 5 train steps at batch 8, E=128 (after 1 untimed), {host}, {t1:.1f}s"}}


def torch_yardstick(lit, batch):
    """CHECKER (oracle/cvcl_oracle.py run on torch's own GPU ops, ATen / MIOpen): the oracle's restatement of the reference
    forward on the benchmark's weights and batch in fp32 and under torch.autocast(bfloat16) -- what the reference does under
    Lightning's --precision bf16.  -> (autocast-vs-fp32 deviations, the torch-fp32 logits).  Not a product path: it pins the
    HIP fp32 mode against torch at the full benchmark size and gives ``logits_rel_vs_fp32`` its yardstick (C2 only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cvcl_oracle as O
    p = {k: v.detach() for k, v in lit.model.state_dict().items()}
    p["logit_neg_log_temperature"] = lit.model.logit_neg_log_temperature.detach().to(batch[0].device).float()
    kw = dict(normalize_features=True, training=True, bn_impl="torch")

    def loss(lpi):
        gt = torch.arange(lpi.shape[0], device=lpi.device)
        lpi = lpi.float()
        return float((torch.nn.functional.cross_entropy(lpi, gt) + torch.nn.functional.cross_entropy(lpi.t(), gt)) / 2)
    with torch.no_grad():
        ref = O.cvcl_forward(p, batch[0], batch[1], batch[2], **kw)[0].float()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = O.cvcl_forward(p, batch[0], batch[1], batch[2], **kw)[0].float()
    a, b = got.double(), ref.double()
    return ({"logits_rel": float((a - b).abs().max() / b.abs().max()),
             "logits_cosine": float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)),
             "loss_abs": abs(loss(got) - loss(ref))}, ref)


def fp8_yardstick(ve, images, patch=16, fp8=True):
    """CHECKER for the ViT configurations (oracle/cvcl_oracle.py::vit_forward on torch's own GPU ops; fp8 = False: bf16 storage points
    only, C4).  C5: what the e4m3 STORAGE POINTS themselves cost --
    the oracle's ViT forward with torch.float8_e4m3fn roundings at the operands of the four linears of every block (per-row scales
    for the LayerNorm outputs and the weights, e8m0 block scales per 32 elements for the attention and GELU outputs) and bf16 at
    the other storage points, against the same forward in fp32 -- next to the HIP e4m3 path against the HIP fp32 mode, both on the
    ViT's output features [B, 768] of the benchmark's weights and frames.  -> dict of rel-L2 / max-rel / cosine for both."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cvcl_oracle as O
    model = ve.model
    p = {k: v.detach().float() for k, v in model.state_dict().items() if not k.startswith("head.")}
    heads = model.blocks[0].attn.num_heads

    def dev(a, b):
        a, b = a.double(), b.double()
        return {"rel_l2": float((a - b).norm() / b.norm()), "max_rel": float((a - b).abs().max() / b.abs().max()),
                "cosine_min": float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())}
    out = {}
    with torch.no_grad():
        chunks = [images[i:i + 64] for i in range(0, images.shape[0], 64)]          # (the fp32 attention matrix of 64 frames: 0.36 GB)
        ref = torch.cat([O.vit_forward(p, c, patch, heads) for c in chunks])
        # (the product folds LayerNorm into the e4m3 qkv / fc1 at this size -- vit_hip.py fold8 -- and the emulation follows it)
        from multimodal import _hip as H
        D = model.embed_dim
        rows = images.shape[0] * ((images.shape[2] // patch) * (images.shape[3] // patch) + 1)
        from multimodal import vit_hip
        fold8 = (bool(fp8) and vit_hip.ln_fold_mode(model) is True and bool(H.lib().cvcl_gemm_fp8_ln_supported(rows, 3 * D, D))
                 and bool(H.lib().cvcl_gemm_fp8_ln_supported(rows, 4 * D, D)))
        emu = torch.cat([O.vit_forward(p, c, patch, heads, quant=O.bf16_round, fp8=fp8, fp8_fold=fold8) for c in chunks])
        out["emulation_vs_torch_fp32"] = dev(emu, ref)
        if fp8:
            out["layernorm_folded"] = fold8
        keep_dt, keep_f8 = model.compute_dtype, getattr(model, "fp8_linears", False)
        try:
            model.compute_dtype, model.fp8_linears = torch.float32, False
            h32 = model(images).float()
            model.compute_dtype, model.fp8_linears = torch.bfloat16, bool(fp8)
            h8 = model(images).float()
        finally:
            model.compute_dtype, model.fp8_linears = keep_dt, keep_f8
        out["hip_fp8_vs_hip_fp32" if fp8 else "hip_bf16_vs_hip_fp32"] = dev(h8, h32)
        out["hip_fp32_vs_torch_fp32"] = dev(h32, ref)
    return {k: ({kk: float(f"{vv:.4g}") for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in out.items()}


def spawn_ranks(a, argv):
    """``python bench.py --gpus N`` outside torch.distributed.run: start N fresh ranks as a child process.  Nothing in this
    process has initialised the GPU at this point (torch.cuda.device_count() does not), and the child is a child -- no exec."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def dist_selfcheck(device, world, rank, local_rank):
    """What the multi-rank line carries so that the first 8-GPU run verifies itself: the backend torch.distributed resolved, how many
    ranks an all-reduce of ones saw, which physical device every rank sits on (two ranks on one device under RCCL = a mis-launch:
    fail loudly), and the two collectives of the path timed on their own -- all-gather of the normalised features (2 x [256, 512]
    f32 per rank) and all-reduce of the frozen configuration's gradient bucket (9 MB) -- with HIP events around collective + wait."""
    backend = dist.get_backend()
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    props = torch.cuda.get_device_properties(device)
    ident = str(getattr(props, "uuid", "")) or f"{props.name}#{getattr(props, 'pci_bus_id', device.index)}"
    mine = {"rank": rank, "local_rank": local_rank, "device_index": device.index, "device": ident}
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    dup = len({(d["device"], d["device_index"]) for d in seen}) < world
    if dup and backend == "nccl":
        raise SystemExit(f"bench.py: two ranks resolved to one device under RCCL: {seen}")

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize(device)
        t = torch.tensor([a.elapsed_time(b) / n * 1e3], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(float(t.item()), 1)
    feat = torch.randn(2 * PER_GPU_BATCH, EMBEDDING_DIM, device=device)
    gathered = torch.empty(world * 2 * PER_GPU_BATCH, EMBEDDING_DIM, device=device)
    bucket = torch.randn(9 * 1024 * 1024 // 4, device=device)
    out = {"backend": backend, "ranks_seen": int(ones.item()), "ranks": seen, "shared_devices": dup,
           "allgather_us": timed(lambda: dist.all_gather_into_tensor(gathered, feat)),
           "allreduce_us": timed(lambda: dist.all_reduce(bucket)),
           "collectives_note": "max over ranks of the HIP-event time per call (collective + the current stream's wait for it), 20 calls "
                               "back to back: all-gather of 2 x [256, 512] f32 features per rank; all-reduce of a 9 MB f32 gradient bucket"}
    if out["ranks_seen"] != world:
        raise SystemExit(f"bench.py: an all-reduce of ones over {world} ranks returned {out['ranks_seen']}")
    return out


# whole-step algorithmic work per pair: forward flops (2 x MAC) of the frozen trunk and, for C2, the bytes
# of a perfectly fused bf16 forward (0.3 MB input + 2 x 28.8 MB activations)
STEP_FLOPS_PER_PAIR = {"c2": 8.46e9, "c4": 35.1e9, "c5": 35.1e9, "c4p14": 46.3e9}
STEP_BYTES_PER_PAIR = {"c2": 58e6}
GEMM_CLASSES = ("gemm", "gemm8w", "gemm_pro")          # the bf16 MFMA GEMM kernels: 128 x 128 glds / 8-wave 256 x 256 / BN-prologue
KERNEL_OF_CLASS = {"gemm": "gemm_glds_kernel", "gemm8w": "gemm8w_kernel", "gemm_pro": "gemm_pro_kernel"}


def static_traffic(kernel_name):
    """PMC L2 <-> fabric bytes per launch of one kernel from the tracked summary of separate rocprofv3 --pmc passes of this
    command (tools/pmc_bench.sh + tools/pmc_summary.py; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).  STATIC: collected on an
    earlier box of this round, not in the run that prints the line.  -> (bytes per launch | None, source)."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_traffic.json")
        try:
            with open(path) as f:
                pm = json.load(f)[kernel_name]
            return (int((pm["hbm_read_bytes_per_step_corrected_x2"] + pm["hbm_write_bytes_per_step"]) / pm["launches_per_step"]),
                    f"profiles/{rnd}_pmc_hbm_traffic.json (static: separate rocprofv3 --pmc passes on an earlier box, tools/pmc_bench.sh)")
        except Exception:
            continue
    return None, None


def static_traffic_vit(cfg):
    """The same for the ViT configurations: HBM bytes per launch averaged over the trunk's GEMM launches (the bf16 8-wave kernel, or
    the two e4m3 kernels), from the tracked per-kernel summary of tools/pmc_cfg.sh <cfg> (FETCH_SIZE x 2 + WRITE_SIZE; STATIC).
    -> (bytes per launch | None, source)."""
    for rnd in (PROFILE_ROUND, "r05", "r04"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{cfg}_summary.json")
        try:
            with open(path) as f:
                pm = json.load(f)
            n = tot = 0.0
            for k, v in pm.items():
                if k.startswith(("gemm8w_kernel", "gemm8f_kernel", "gemm_fp8_kernel")) and "hbm_read_MB_x2" in v:
                    n += v["dispatches_per_pass"]
                    tot += (v["hbm_read_MB_x2"] + v["hbm_write_MB"]) * 1e6 * v["dispatches_per_pass"]
            if n:
                return int(tot / n), (f"profiles/{rnd}_pmc_{cfg}_summary.json (static: separate rocprofv3 --pmc passes on an earlier "
                                      "box, tools/pmc_cfg.sh; mean over the trunk's GEMM launches)")
        except Exception:
            pass
    return None, None


def measure(cfg, precision, batch_size, steps, warmup, device, world, rank, *, roofline=True, parity=True, yardstick=False,
            seed=None):
    """One benchmark configuration end to end: build the module, (parity check), warm up, time ``steps`` steps between
    barriers, (instrumented passes for the roofline).  -> dict of results (rank-local; ``elapsed`` is the max over ranks)."""
    from multimodal import _hip as H
    from multimodal import parallel
    lit, ve, opt = build_model(cfg, device, precision)
    engine = parallel.DataParallelEngine(device, global_negatives=True).attach(lit)
    batch = synthetic_batch_on_device(batch_size, seed=rank if seed is None else seed, device=device) + (None,)

    # The frozen trunk runs on its own HIP stream(s) (H.TrunkStream): step k+1's trunk overlaps step k's trainable tail -- head,
    # text encoder, loss, backward, AdamW, and with world > 1 the feature all-gathers, the global-negatives loss and the deferred
    # all-reduce wait + optimizer step -- which stays on the main stream; with two trunk streams consecutive passes of the frozen
    # trunk also overlap each other (BatchNorm running statistics still updated in step order).  Every step does the same work
    # with the same numbers (bit-identical, tests/test_train_entry_gpu.py), all of it complete when the clock stops.  The number
    # of trunk streams is fixed up front -- $CVCL_TRUNK_STREAMS (ResNeXt) / $CVCL_VIT_TRUNK_STREAMS (ViT), default 2, 0 = the
    # single-stream schedule -- and reported in config.trunk_streams; nothing is auto-selected at run time.
    env_name = "CVCL_TRUNK_STREAMS" if cfg == "c2" else "CVCL_VIT_TRUNK_STREAMS"
    trunk_streams = int(os.environ.get(env_name, "2" if batch_size <= 1024 else "1"))     # (large batches: one pass's scratch at a time)
    if os.environ.get("CVCL_TRUNK_STREAM", "1") == "0":
        trunk_streams = 0
    torch.cuda.synchronize()

    def set_trunk_streams(n):
        if cfg == "c2":
            ve.model.enable_trunk_stream(device, inputs="ready" if n else None, n_streams=max(n, 1))
        else:
            from multimodal import vit_hip
            vit_hip.enable_trunk_stream(ve.model, device, inputs="ready" if n else None, n_streams=max(n, 1))
    set_trunk_streams(trunk_streams)              # the benchmark batch is resident and never rewritten: inputs="ready"

    # multi-GPU: the all-reduce + optimizer step of step k are enqueued behind the frozen trunk of step k+1
    # (parallel.OverlappedUpdate; same parameter sequence as the sequential schedule); flushed before the clock stops
    upd = parallel.OverlappedUpdate(engine, opt, ve) if world > 1 else None

    def step():
        if upd is None:
            opt.zero_grad(set_to_none=True)
            out = lit.training_step(batch, 0)
            out["loss"].backward()
            engine.reduce_gradients()
            opt.step()
            return out
        out = lit.training_step(batch, 0)          # trunk, then (hook) the previous step's update, then fc / text / loss
        upd.zero_grad()
        out["loss"].backward()
        upd.step_done()
        return out

    def flush():
        if upd is not None:
            upd.flush()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = {"config": cfg, "precision": precision, "batch": batch_size, "steps": steps, "warmup": warmup, "trunk_streams": trunk_streams}

    # Parity of the benchmarked precision, on the benchmark's weights (random init, before any optimizer step) and batch,
    # train-mode BatchNorm; BatchNorm buffers restored, so the timed steps start from the initial state:
    #  * vs the exact-fp32 parity mode of the same kernels (logits_rel_vs_fp32 ...);
    #  * C2: the fp32 mode itself vs torch's own fp32 ops running the oracle's forward (the 1e-3 gate at the full size), and
    #    what torch's own autocast(bf16) does on the same weights and batch -- the yardstick for the bf16 figure.
    # The ResNeXt trunk stores its raw convolution outputs centred on calibrated batch means (resnext.py): the calibration is
    # done on ANOTHER batch of the same distribution first, as in a training run (the calibration batch is not the evaluated one).
    if parity and batch_size <= 1024:
        set_trunk_streams(0)
        par = {}
        if precision != "32":
            if cfg == "c2" and hasattr(ve.model, "recalibrate_centres"):
                keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
                other = synthetic_batch_on_device(batch_size, seed=977 + rank, device=device)
                gn, lit.model.global_negatives = lit.model.global_negatives, False      # (rank-local: no collective in this check)
                with torch.no_grad():
                    lit.model(other[0], other[1], other[2])
                lit.model.global_negatives = gn
                lit.load_state_dict(keep, strict=False)
            par.update(logits_vs_fp32(lit, batch, precision))
        if yardstick and cfg == "c2":
            ty, torch_logits = torch_yardstick(lit, batch)
            lit.set_precision("32")
            te_training = lit.model.text_embed.training
            lit.model.text_embed.eval()
            keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
            gn, lit.model.global_negatives = lit.model.global_negatives, False
            with torch.no_grad():
                li, _ = lit.model(batch[0], batch[1], batch[2])
            lit.model.global_negatives = gn
            lit.load_state_dict(keep, strict=False)
            lit.model.text_embed.train(te_training)
            lit.set_precision(precision)
            a_, b_ = li.double(), torch_logits.double()
            par["hip_fp32_logits_rel_vs_torch_fp32"] = float((a_ - b_).abs().max() / b_.abs().max())
            par["torch_autocast_bf16_vs_torch_fp32"] = {k: float(f"{v:.4g}") for k, v in ty.items()}
            if precision == "bf16":
                par["conditioned"] = structured_parity(lit, ve, batch_size, device, rank)
        if cfg == "c5" and precision == "fp8":
            par["fp8_features_yardstick"] = fp8_yardstick(ve, batch[0], patch_of(cfg))
        elif cfg in ("c4", "c4p14") and precision == "bf16":                # the same with bf16 storage points only
            par["bf16_features_yardstick"] = fp8_yardstick(ve, batch[0], patch_of(cfg), fp8=False)
        torch.cuda.synchronize()
        set_trunk_streams(trunk_streams)
        res["parity"] = par

    for _ in range(warmup):
        out = step()
    flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        per_rank = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([elapsed], device=device, dtype=torch.float64))
        per_rank = [float(t.item()) for t in per_rank]
        res["ms_per_step_per_rank"] = {"min": round(min(per_rank) / steps * 1e3, 3), "max": round(max(per_rank) / steps * 1e3, 3)}
        elapsed = max(per_rank)
    res["elapsed"] = elapsed
    res["final_loss"] = float(out["loss"].detach())
    res["value"] = world * batch_size * steps / elapsed
    res["ms_per_step"] = elapsed / steps * 1e3
    pairs_s_gpu = batch_size * steps / elapsed
    peak_tf = MFMA_FP8_PEAK_TFLOPS if precision == "fp8" else MFMA_BF16_PEAK_TFLOPS
    res["whole_step"] = {"mfma_frac": round(pairs_s_gpu * STEP_FLOPS_PER_PAIR[cfg] / 1e12 / peak_tf, 4),
                         "flops_per_pair": STEP_FLOPS_PER_PAIR[cfg], "peak_tflops": peak_tf,
                         "note": "per-GPU pairs/s x the frozen trunk's forward flops per pair (SURVEY.md 8d) / the dense MFMA peak of the dtype"}
    if cfg in STEP_BYTES_PER_PAIR and precision == "bf16":
        res["whole_step"].update({"hbm_frac": round(pairs_s_gpu * STEP_BYTES_PER_PAIR[cfg] / 1e9 / HBM_PEAK_GBS, 4),
                                  "bytes_per_pair": STEP_BYTES_PER_PAIR[cfg]})
        if batch_size == PER_GPU_BATCH:
            # the bytes the step ACTUALLY moves (PMC, static summary of the same command at this batch) against the same clock:
            # what fraction of the HBM peak the whole step sustains on its real traffic
            for rnd in (PROFILE_ROUND, "r05", "r04"):
                try:
                    with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_traffic.json")) as f:
                        gb = json.load(f)["trunk_total"]["total_GB"]
                    res["whole_step"].update({"pmc_traffic_gb_per_step": round(gb, 2),
                                              "pmc_traffic_frac_of_hbm_peak": round(gb / (elapsed / steps) / HBM_PEAK_GBS, 4),
                                              "pmc_traffic_source": f"profiles/{rnd}_pmc_hbm_traffic.json (static)"})
                    break
                except Exception:
                    pass

    if roofline and precision != "32":
        # further passes of the same steps with HIP events around every launch (on the launch stream).  The roofline figures come
        # from a pass with ONE trunk pass in flight: with two trunk streams a launch shares the CUs and HBM with the other pass's
        # kernels and its event-timed duration is no longer the kernel's own (the rocprofv3 --kernel-trace run of this command
        # reports the one-at-a-time durations too); the durations seen under the two-stream schedule are reported beside them.
        def instrumented(n):
            for _ in range(2):
                step()
            flush()
            torch.cuda.synchronize()
            H.prof_enable(True)
            for _ in range(n):
                step()
            flush()
            torch.cuda.synchronize()
            out_ = H.prof_collect()
            H.prof_enable(False)
            return out_

        nprof, nconc, conc = min(steps, 10), min(steps, 5), None
        if trunk_streams > 1:
            conc = instrumented(nconc)
            set_trunk_streams(1)
        prof = instrumented(nprof)
        # An event bracket (previous launch's end -> this launch's end on the stream) also holds the dispatch gap ahead of the
        # kernel.  The gap is calibrated per run with a kernel that does nothing (cvcl_prof_null_bracket_us: two event packets +
        # one dispatch + ~3.6 us of empty kernel) and taken per launch as HALF of that bracket for launch-bound classes (average
        # bracket < 4 null brackets: bn_finalize, head) and a QUARTER of it for long kernels (the command processor has the
        # packets decoded while the previous kernel still runs) -- a two-point calibration against rocprofv3 --kernel-trace of
        # this command: bn_finalize 4.8 vs 5.1 us, gemm8w 76.7 vs 77.1, gemm_pro 107.5 vs 109.0, gconv 57.6 vs 56.5
        # (profiles/r03_bench_c2_1stream_kernel_stats.csv).  event_ms = bracket sums, kernel_ms = event_ms - gap_ms.
        null_us = H.prof_null_bracket_us()

        def gap_us_of(c):
            avg_bracket = prof[c][0] / max(prof[c][1], 1) * 1e3
            return null_us * (0.25 if avg_bracket > 4 * null_us else 0.5)
        gap = {k: v[1] / nprof * gap_us_of(k) * 1e-3 for k, v in prof.items() if v[1] > 0}
        res["event_ms_per_step"] = {k: round(v[0] / nprof, 4) for k, v in prof.items() if v[1] > 0}
        res["kernel_ms_per_step"] = {k: round(max(v[0] / nprof - gap[k], 0.0), 4) for k, v in prof.items() if v[1] > 0}
        res["gap_ms_per_step"] = {k: round(g, 4) for k, g in gap.items()}
        res["launches_per_step"] = {k: v[1] // nprof for k, v in prof.items() if v[1] > 0}
        res["event_bracket_of_a_null_kernel_us"] = round(null_us, 2)

        def kern_ms(c):                                         # per step, gap removed
            return max(prof[c][0] / nprof - gap.get(c, 0.0), 1e-9)
        dom = max((c for c in prof if prof[c][1] > 0), key=kern_ms)
        per_kernel = None
        if cfg == "c2":
            per_kernel, nbytes, flops, launches = resnext_gemm_work(batch_size)
        else:
            nbytes, flops, launches = vit_gemm_work(batch_size, patch=patch_of(cfg), operand_bytes=1 if precision == "fp8" else 2)
        by = {}
        for c in GEMM_CLASSES:
            if not prof[c][1]:
                continue
            v = {"kernel_ms_per_step": round(kern_ms(c), 4), "launches_per_step": prof[c][1] // nprof}
            pb, pf, pn = per_kernel[c] if per_kernel is not None else (0, 0, -1)
            if pn == v["launches_per_step"]:
                ms = kern_ms(c)
                v.update({"avg_launch_us": round(ms * 1e3 / pn, 2), "algorithmic_bytes_per_launch": int(pb / pn),
                          "algorithmic_flops_per_launch": int(pf / pn), "algorithmic_GBps": round(pb / ms / 1e6, 1),
                          "tflops": round(pf / ms / 1e9, 1), "hbm_frac": round(pb / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "mfma_frac": round(pf / ms / 1e9 / peak_tf, 4)})
            by[c] = v
        g_ms = sum(kern_ms(c) for c in GEMM_CLASSES if prof[c][1])
        g_n = sum(prof[c][1] for c in GEMM_CLASSES) // nprof
        family = {"launches_per_step": g_n, "kernel_ms_per_step": round(g_ms, 4), "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2),
                  "algorithmic_GBps": round(nbytes / g_ms / 1e6, 1), "tflops": round(flops / g_ms / 1e9, 1),
                  "hbm_frac": round(nbytes / g_ms / 1e6 / HBM_PEAK_GBS, 4), "mfma_frac": round(flops / g_ms / 1e9 / peak_tf, 4)}
        # the record's headline = the DOMINANT kernel class by time against its own algorithmic work
        d = by.get(dom)
        if cfg != "c2":
            # ViT configurations: the trunk's 49 GEMM launches (patch embedding + 4 linears x 12 blocks) are ONE kernel -- the bf16
            # linear-epilogue gemm8w, or the e4m3 gemm_fp8 (profiled under the class "gemm") -- so the family figure is the kernel's
            rl = {"kernel": ("gemm8f_kernel / gemm_fp8_kernel (e4m3 x e4m3 ViT linears on v_mfma_scale_f32_32x32x64_f8f6f4: 8-wave 256|192 x 256 "
                              "tiles for qkv / fc1, 128 x 128 tiles for proj / fc2)" if precision == "fp8"
                             else "gemm8w_kernel<linear epilogue> (bf16 ViT linears: bias / GELU / residual; 8-wave 256|224 x 256 tiles)"),
                  "dominant_class_by_time": dom, "bound": "mfma", "achieved": family["tflops"], "peak": peak_tf, "unit": "TFLOP/s",
                  "frac": family["mfma_frac"], "traffic": static_traffic_vit(cfg)[0], "traffic_source": static_traffic_vit(cfg)[1],
                  "traffic_note": ("L2 <-> fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE), Infinity-Cache hits included: a 4 MB L2 holds the 8 m-tiles of a "
                                   "supertile's A block OR the W matrix (3.5-4.7 MB), not both, so W is re-read once per super-row (e.g. fc1: "
                                   "28 x 4.7 MB on top of 77 MB of A) -- served by the 256 MB Infinity Cache, not by HBM"),
                  "other_bound_frac": family["hbm_frac"],
                  "avg_launch_us": family["avg_launch_us"], "launches_per_step": family["launches_per_step"],
                  "algorithmic_bytes_per_launch": int(nbytes / launches), "algorithmic_flops_per_launch": int(flops / launches)}
        elif d is not None and "tflops" in d:
            # which roofline bounds it: arithmetic intensity against the ridge (peak flops / peak bytes)
            intensity = d["algorithmic_flops_per_launch"] / d["algorithmic_bytes_per_launch"]
            ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
            mf = intensity >= ridge or cfg != "c2"
            traffic, src = static_traffic(KERNEL_OF_CLASS[dom]) if cfg == "c2" else (None, None)
            rl = {"kernel": KERNEL_OF_CLASS[dom] if cfg == "c2" else
                  ("gemm_fp8_kernel (e4m3 x e4m3 ViT linears on v_mfma_scale_f32_32x32x64_f8f6f4)" if precision == "fp8"
                   else "gemm8w_kernel<linear epilogue> (bf16 ViT linears: bias / GELU / residual; 8-wave 256|224 x 256 tiles)"),
                  "dominant_class_by_time": dom,
                  "bound": "mfma" if mf else "hbm",
                  "achieved": d["tflops"] if mf else d["algorithmic_GBps"], "peak": peak_tf if mf else HBM_PEAK_GBS,
                  "unit": "TFLOP/s" if mf else "GB/s", "frac": d["mfma_frac"] if mf else d["hbm_frac"],
                  "traffic": traffic, "traffic_source": src,
                  "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                  "other_bound_frac": d["hbm_frac"] if mf else d["mfma_frac"],
                  "avg_launch_us": d["avg_launch_us"], "launches_per_step": d["launches_per_step"],
                  "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                  "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"]}
        else:                                                   # the dominant class is not a GEMM (or launch lists disagree): the family
            rl = {"kernel": "conv/linear GEMM family", "dominant_class_by_time": dom, "bound": "hbm" if cfg == "c2" else "mfma",
                  "achieved": family["algorithmic_GBps"] if cfg == "c2" else family["tflops"],
                  "peak": HBM_PEAK_GBS if cfg == "c2" else peak_tf, "unit": "GB/s" if cfg == "c2" else "TFLOP/s",
                  "frac": family["hbm_frac"] if cfg == "c2" else family["mfma_frac"], "traffic": None}
        rl["by_kernel"] = by
        rl["gemm_family_blend"] = family
        rl["timing"] = ("HIP events on the launch stream around every launch (cvcl_prof_enable), one trunk pass in flight, minus the "
                        "event bracket of a null kernel per launch = the kernel's own duration, which is what rocprofv3 --kernel-trace of "
                        "this command reports (profiles/)")
        if conc is not None:
            c_ms = sum(conc[c][0] for c in GEMM_CLASSES)
            c_n = sum(conc[c][1] for c in GEMM_CLASSES)
            rl["concurrent"] = {"note": "the timed region keeps two trunk passes in flight on two HIP streams; event-timed there, a "
                                        "launch's duration includes the time it shares the GPU with the other pass's kernels -- "
                                        "per-kernel figures are not meaningful, the step time is",
                                "avg_launch_us": round(c_ms / max(c_n, 1) * 1e3, 2)}
        res["roofline"] = rl
        if world > 1:
            dist.barrier()
    res["hbm_peak_gb"] = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)
    set_trunk_streams(0)
    torch.cuda.synchronize()
    if world == 1 and precision != "32" and batch_size <= 1024:
        # the trainable TAIL on its own (round 5): the frozen trunk replaced by its cached output, nothing else on the GPU -- fc / head,
        # text encoder forward + backward, L2 normalise, logits, InfoNCE forward + backward, AdamW.  Event-timed per step; launches =
        # every device activity torch's profiler sees in a step (library kernels, torch's own kernels, device-to-device copies)
        try:
            with frozen_trunk_cached(ve, cfg, batch[0]):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n_t = 10
                e0.record()
                for _ in range(n_t):
                    step()
                e1.record()
                torch.cuda.synchronize()
                res["tail_ms_per_step"] = round(e0.elapsed_time(e1) / n_t, 4)
                H.prof_enable(True)
                for _ in range(n_t):
                    step()
                torch.cuda.synchronize()
                tp = H.prof_collect()
                H.prof_enable(False)
                res["tail_library_launches_per_step"] = int(sum(v[1] for v in tp.values()) // n_t)
                try:
                    from torch.profiler import ProfilerActivity, profile
                    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof_:
                        for _ in range(3):
                            step()
                        torch.cuda.synchronize()
                    n_dev = sum(1 for ev in prof_.events() if str(ev.device_type).endswith("CUDA"))
                    res["tail_launches_per_step"] = round(n_dev / 3, 1) if n_dev else None
                except Exception:
                    res["tail_launches_per_step"] = None
        except Exception as e:                                  # the side measurement must never take the bench line down
            res["tail_error"] = repr(e)[:200]
        torch.cuda.synchronize()
    del lit, ve, opt, engine, batch, upd
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def measure_finetune(device, batch_size, steps=5, warmup=3):
    """C2 with --finetune_cnn (SURVEY 8d "finetune" mode; the reference's *_finetune_cnn runner configs): the whole ResNeXt-50 trains --
    forward with saved activations, BatchNorm / conv data and weight gradients (multimodal/trunk_train.py), AdamW over all 161 trunk
    parameters.  Single stream, bf16 storage.  -> sub-record."""
    import multimodal.multimodal as mm
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel
    torch.manual_seed(0)
    args = c2_args()
    args.finetune_cnn = True
    with contextlib.redirect_stdout(io.StringIO()):
        ve = VisionEncoder(args)
        te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
        lit = MultiModalLitModel(ve, te, args)
    lit.to(device)
    lit.set_precision("bf16")
    lit.train()
    opt = lit.configure_optimizers()
    opt = opt["optimizer"] if isinstance(opt, dict) else opt
    batch = synthetic_batch_on_device(batch_size, 0, device) + (None,)

    def step():
        opt.zero_grad(set_to_none=True)
        out = lit.training_step(batch, 0)
        out["loss"].backward()
        opt.step()
        return out
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_train = sum(p.numel() for p in lit.parameters() if p.requires_grad)
    return {"value": round(batch_size / dt, 1), "unit": "pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup,
            "dtype": "bf16", "per_gpu_batch": batch_size, "trainable_parameters": n_train, "final_loss": round(float(out["loss"]), 5),
            "note": "C2 with --finetune_cnn: forward + backward through the whole ResNeXt-50 (BatchNorm train mode) + AdamW; the frozen "
                    "default is the headline above"}


def sub_record(r):
    """The bounded side measurements of the default line (other precisions / configurations of BASELINE.json)."""
    out = {"value": round(r["value"], 1), "unit": "pairs/s", "ms_per_step": round(r["ms_per_step"], 3), "steps": r["steps"],
           "warmup": r["warmup"], "per_gpu_batch": r["batch"], "dtype": {"bf16": "bf16", "32": "f32", "fp8": "fp8-e4m3"}[r["precision"]],
           "workload": WORKLOADS[r["config"]], "whole_step": r["whole_step"], "final_loss": round(r["final_loss"], 5)}
    if r.get("parity"):
        out["parity"] = {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in r["parity"].items()}
    if r.get("roofline"):
        out["roofline"] = {k: v for k, v in r["roofline"].items() if k not in ("by_kernel", "timing", "concurrent")}
        out["kernel_ms_per_step"] = r["kernel_ms_per_step"]
    for k in ("tail_ms_per_step", "tail_launches_per_step", "tail_library_launches_per_step", "tail_error"):
        if k in r:
            out[k] = r[k]
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5", "c4p14"])
    ap.add_argument("--batch", default=None, help="per-GPU batch: a number, or 'auto' = as many pairs as fit the free HBM (frozen-ViT "
                                                  "configurations: ~5 MB per pair, capped at 16384); default 256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bounded sub-records of the default C2 line (fp32 parity mode, C4, C5)")
    ap.add_argument("--precision", default=None, choices=["bf16", "32", "fp8"])
    a = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a, argv))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the CVCL hot path has no CPU fallback")
    backend = os.environ.get("CVCL_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo lets two ranks share one GPU (smoke test of the N > 1 path)
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {ndev} visible device(s): one rank per GPU under RCCL "
                         "(CVCL_DIST_BACKEND=gloo shares a device for smoke tests)")
    device = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(device)
    selfcheck = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
        selfcheck = dist_selfcheck(device, world, rank, local_rank)

    cfg = a.config
    precision = a.precision or ("fp8" if cfg == "c5" else "bf16")
    if a.batch in (None, ""):
        batch_size = PER_GPU_BATCH
    elif str(a.batch) == "auto":
        # BASELINE configs[4]: "per-GPU batch sized to 288 GB HBM".  Measured on MI355X (C5): 5.1 MB of HBM per pair (fp32 frame, patch
        # matrix, tokens, qkv, MLP hidden, e4m3 copies, two trunk passes in flight) + 3 x B^2 x 4 bytes of logits / gradients in the
        # head: 2.0 GB at B 256, 11.0 at 2048, 41.8 at 8192.  80 % of the free memory, multiples of 1024, at most 16384 pairs (the
        # largest batch the head kernels have been run at).  Throughput does NOT grow with the batch -- 27.8 k pairs/s at 256,
        # 26.9 k at 2048, 26.4 k at 8192, 25.0 k at 16384: the trunk GEMMs already have M = 50 k rows at B 256 and the InfoNCE
        # head is O(B^2) -- so the default stays 256 and 'auto' exists to show the configuration runs at HBM scale.
        free_b, _tot = torch.cuda.mem_get_info(device)
        batch_size = 1024
        while batch_size + 1024 <= 16384 and (batch_size + 1024) * 5.1e6 + 12.0 * (batch_size + 1024) ** 2 < 0.8 * free_b:
            batch_size += 1024
    else:
        batch_size = int(a.batch)
    steps = a.steps if a.steps is not None else (50 if batch_size <= 1024 else 5)
    warmup = a.warmup if a.warmup is not None else (10 if batch_size <= 1024 else 2)
    default_line = cfg == "c2" and a.precision is None and a.batch in (None, "") and world == 1
    r = measure(cfg, precision, batch_size, steps, warmup, device, world, rank, roofline=not a.no_roofline, parity=not a.no_parity,
                yardstick=default_line and not a.no_parity)

    extras = {}
    if default_line and not a.no_extras:
        # bounded sub-records (>= 10 timed steps each at 256 pairs): the C2 step in the mode that meets the 1e-3 logits gate
        # (exact-fp32 MFMA / fp32 storage), and BASELINE configs[3] / [4] on one GPU, each with its own roofline and parity
        sub_steps = max(10, min(steps, 20))
        r32 = measure("c2", "32", PER_GPU_BATCH, 10, 3, device, world, rank, roofline=False, parity=False)
        extras["fp32_parity_mode"] = sub_record(r32)
        extras["fp32_parity_mode"]["note"] = ("C2 with --precision 32: the mode held to the 1e-3 logits gate (2e-5 vs the reference's "
                                              "golden logits; parity.hip_fp32_logits_rel_vs_torch_fp32 above is this mode at the full size)")
        for c in ("c4", "c5", "c4p14"):
            extras[c] = sub_record(measure(c, "fp8" if c == "c5" else "bf16", PER_GPU_BATCH, sub_steps, 5, device, world, rank,
                                           roofline=not a.no_roofline, parity=not a.no_parity))
        extras["finetune_cnn"] = measure_finetune(device, PER_GPU_BATCH)

    if rank == 0:
        line = {"metric": METRIC if cfg == "c2" else f"image-text pairs/sec, CVCL ViT-B/{patch_of(cfg)}+transformer text 224², {cfg.upper()}, MI355X",
                "value": round(r["value"], 1), "unit": "pairs/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "32": "f32", "fp8": "fp8-e4m3"}[precision],
                "data": "synthetic",
                "config": {"workload": WORKLOADS[cfg], "per_gpu_batch": batch_size, "global_batch": batch_size * world,
                           "negatives": "global (RCCL all-gather)" if world > 1 else "local (single GPU)",
                           "parallelism": f"dp{world}", "trunk_streams": r["trunk_streams"]},
                "final_loss": round(r["final_loss"], 5), "hbm_peak_gb": r["hbm_peak_gb"], "whole_step": r["whole_step"]}
        if selfcheck is not None:
            line["distributed"] = dict(selfcheck, ms_per_step_per_rank=r.get("ms_per_step_per_rank"))
        par = r.get("parity")
        if par:
            line.update({k: float(f"{v:.4g}") for k, v in par.items() if isinstance(v, float)})
            line["parity"] = {k: v for k, v in par.items() if not isinstance(v, float)}
            line["parity_note"] = ("the benchmark's random-init weights and batch, train-mode BatchNorm, before the first optimizer step; "
                                   "logits_rel = max |d logit| / max |logit|.  *_vs_fp32: the benchmarked precision against the exact-fp32 "
                                   "parity mode of the same kernels; hip_fp32_logits_rel_vs_torch_fp32: that fp32 mode against torch's own "
                                   "fp32 ops running the oracle's forward (the 1e-3 gate, at the full benchmark size); "
                                   "torch_autocast_bf16_vs_torch_fp32: what torch's own bf16 autocast does on the same weights and batch "
                                   "(checker = oracle/ on torch GPU ops).  iid-noise frames through a random-init trunk amplify EVERY bf16 "
                                   "rounding ~100x (weights, inputs, activations alike): DESIGN.md section 3")
        if r.get("roofline"):
            line["roofline"] = r["roofline"]
            line["kernel_ms_per_step"] = r["kernel_ms_per_step"]
            line["event_ms_per_step"] = r["event_ms_per_step"]
            line["gap_ms_per_step"] = r["gap_ms_per_step"]
            line["launches_per_step"] = r["launches_per_step"]
            line["event_bracket_of_a_null_kernel_us"] = r["event_bracket_of_a_null_kernel_us"]
        for k in ("tail_ms_per_step", "tail_launches_per_step", "tail_library_launches_per_step", "tail_error"):
            if k in r:
                line[k] = r[k]
        if "tail_ms_per_step" in r:
            line["tail_note"] = ("the step with the frozen trunk replaced by its cached output (fc / head, text encoder, loss, backward, "
                                 "AdamW), nothing else on the GPU; tail_launches_per_step = device activities torch.profiler sees per step")
        line.update(extras)
        if world == 1 and not a.no_cpu_baseline and cfg == "c2":
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
