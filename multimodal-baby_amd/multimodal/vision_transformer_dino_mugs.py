"""DINO ViT with the reference's module tree / state_dict keys (reference
multimodal/vision_transformer_dino_mugs.py:87-299), computed by libcvcl_hip.

``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Conv2d`` children are parameter containers (timm/DINO names and
init: trunc-normal 0.02 weights, zero biases, unit LayerNorm); ``forward`` runs the HIP kernels."""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from . import _hip as H


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("dropout / stochastic depth are 0 in every CVCL configuration")
        self.num_features = self.embed_dim = embed_dim
        self.num_heads, self.patch_size = num_heads, patch_size
        self.patch_embed = PatchEmbed(img_size=img_size[0], patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                           qk_scale=qk_scale, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.compute_dtype = torch.float32
        self._cache = {}
        self._pre_head_callback = None                  # parallel.OverlappedUpdate: runs between the trunk and the head

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_cache"] = {}
        d["_pre_head_callback"] = None
        d.pop("_trunk_stream", None)          # streams / ring buffers of H.TrunkStream are never pickled
        d.pop("_trunk_out", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.__dict__.setdefault("compute_dtype", torch.float32)
        self.__dict__.setdefault("_cache", {})
        self.__dict__.setdefault("_pre_head_callback", None)
        # instances pickled by the reference's own class (hyper_parameters of its Lightning checkpoints) carry only what
        # reference :174-197 sets: recover the two attributes the HIP forward reads from the submodules
        mods = self.__dict__.get("_modules", {})
        if "patch_size" not in self.__dict__ and "patch_embed" in mods:
            self.__dict__["patch_size"] = int(mods["patch_embed"].patch_size)
        if "num_heads" not in self.__dict__ and len(mods.get("blocks", ())) > 0:
            self.__dict__["num_heads"] = int(mods["blocks"][0].attn.num_heads)

    def enable_trunk_stream(self, device, inputs="caller", stream=None, n_streams=None):
        """Run the frozen ViT on a stream of its own so it overlaps the previous step's trainable tail (see H.TrunkStream)."""
        from . import vit_hip
        return vit_hip.enable_trunk_stream(self, device, inputs, stream, n_streams)

    def interpolate_pos_encoding(self, x, w, h):
        """Position table for an input whose patch grid differs from the one the table was learned on (reference :210-230; same
        arguments: ``x`` = the token tensor -- only its token count and width are read --, ``w`` / ``h`` = the image's dim 2 / dim 3).
        Identity at the native square resolution; otherwise the patch part is resampled bicubically to (w // patch, h // patch) with the
        reference's +0.1 on the target sizes (its guard against an output one short), the class-token entry kept.  A once-per-shape
        parameter transform (cached by the caller), not hot-path arithmetic: it runs on ``F.interpolate``."""
        n_native = self.pos_embed.shape[1] - 1
        if x.shape[1] - 1 == n_native and w == h:
            return self.pos_embed
        side = int(math.sqrt(n_native))
        gh, gw = w // self.patch_size, h // self.patch_size
        table = self.pos_embed[:, 1:].reshape(1, side, side, -1).permute(0, 3, 1, 2)
        table = nn.functional.interpolate(table, scale_factor=((gh + 0.1) / math.sqrt(n_native), (gw + 0.1) / math.sqrt(n_native)),
                                          mode="bicubic")
        if table.shape[-2] != gh or table.shape[-1] != gw:
            raise AssertionError(f"interpolated grid {tuple(table.shape[-2:])} != {(gh, gw)}")
        return torch.cat([self.pos_embed[:, :1], table.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)

    def forward(self, x):
        """-> cls token after the final LayerNorm, [B, D] fp32 (reference :245-250)."""
        from . import vit_hip
        return vit_hip.vit_forward(self, x)


def vit_tiny(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)


def vit_small(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)


def vit_base(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)


def vit_large(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)
