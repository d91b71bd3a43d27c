"""Helpers on the hot path with the reference's names (reference multimodal/utils.py:106-108, 145-214)."""
from __future__ import annotations

import torch

from . import _hip as H


def get_entropy(logits, dim=-1):
    """E[-log p] of softmax(logits) along ``dim`` (reference utils.py:106-108), on the HIP row kernel."""
    if dim not in (-1, logits.dim() - 1):
        logits = logits.transpose(dim, -1)
    shape = logits.shape[:-1]
    x = logits.reshape(-1, logits.shape[-1]).contiguous().float()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    H.check(H.lib().cvcl_row_entropy(H.ptr(x, torch.float32), H.ptr(out), x.shape[0], x.shape[1], H.stream_ptr()),
            "cvcl_row_entropy")
    return out.reshape(shape)


def map_structure(fn, *obj):
    """Apply ``fn`` leaf-wise over (possibly nested) lists / tuples / dicts (reference utils.py:111-138)."""
    first = obj[0]
    if isinstance(first, list):
        return [map_structure(fn, *x) for x in zip(*obj)]
    if isinstance(first, tuple) and not isinstance(first, torch.Size):
        vals = [map_structure(fn, *x) for x in zip(*obj)]
        return type(first)(*vals) if hasattr(first, "_fields") else tuple(vals)
    if isinstance(first, dict):
        return {k: map_structure(fn, *[o[k] for o in obj]) for k in first}
    return fn(*obj)


def apply_permutation(tensor: torch.Tensor, permutation, dim: int) -> torch.Tensor:
    return tensor.index_select(dim, permutation)


_MODEL_SPECS = {"resnext50": ("resnext50_32x4d", None), "vitb14": ("vit_base", 14), "vitl16": ("vit_large", 16),
                "vitb16": ("vit_base", 16), "vits16": ("vit_small", 16)}


def build_dino_mugs(arch, patch_size):
    """Random-init DINO backbone (reference utils.py:199-214): a vendored-ViT factory or ResNeXt-50 with an
    identity fc."""
    from . import resnext
    from . import vision_transformer_dino_mugs as vits
    if arch in vits.__dict__ and callable(vits.__dict__[arch]) and arch.startswith("vit_"):
        return vits.__dict__[arch](patch_size=patch_size, num_classes=0)
    if arch == "resnext50_32x4d":
        model = resnext.resnext50_32x4d()
        model.fc = torch.nn.Identity()
        return model
    raise ValueError(f"Unknown architecture: {arch}")


def load_model(model_name, pretrained=True):
    """``alg_data_arch`` -> backbone (reference utils.py:145-178).  The reference downloads the checkpoint from
    the HuggingFace hub before it even looks at ``pretrained`` (Appendix C.7); there is no network here, so
    pretrained weights must be supplied through a local file named by $CVCL_PRETRAINED_DIR/<model_name>.pth."""
    import os
    alg, data, model_spec = model_name.split("_")
    assert alg in ("dino", "mugs"), "Unrecognized algorithm!"
    assert model_spec in _MODEL_SPECS, "Unrecognized architecture!"
    arch, patch = _MODEL_SPECS[model_spec]
    model = build_dino_mugs(arch, patch)
    if pretrained:
        path = os.path.join(os.environ.get("CVCL_PRETRAINED_DIR", ""), model_name + ".pth")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"pretrained weights for {model_name} not found at {path} (no network access: "
                                    "set CVCL_PRETRAINED_DIR, or drop --pretrained_cnn for random init)")
        load_dino_mugs(model, path, "teacher")
    return model


def load_dino_mugs(model, pretrained_weights, checkpoint_key):
    """Load a DINO/Mugs checkpoint, stripping wrapper prefixes (reference utils.py:180-197)."""
    state = torch.load(pretrained_weights, map_location="cpu")
    if checkpoint_key is not None and checkpoint_key in state:
        state = state[checkpoint_key]
    for prefix in ("module.", "backbone.", "encoder."):
        state = {k.replace(prefix, ""): v for k, v in state.items()}
    msg = model.load_state_dict(state, strict=False)
    print(f"Pretrained weights found at {pretrained_weights} and loaded with msg: {msg}")
