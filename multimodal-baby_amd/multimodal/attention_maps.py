"""Forward-hook helper with the reference's ``Hook`` contract (reference multimodal/attention_maps.py:83-105).

Only the hook used inside ``VisionEncoder.forward`` is on the hot path; the Grad-CAM plotting helpers of
the reference file are visualisation code and out of scope.

The public names are the contract callers of the reference rely on and are therefore the same: the constructor
``Hook(module, requires_grad=True)``, use as a context manager, the attributes ``data`` / ``hook`` / ``requires_grad``
and the read-only views ``activation`` / ``gradient``.  The body is this repository's own."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn


class Hook:
    """Captures what ``module`` returns on its next forward call(s); with ``requires_grad`` the captured tensor is made a
    gradient-retaining leaf of the graph that follows, so ``gradient`` is available after a backward pass.  Leaving the
    ``with`` block detaches the hook from the module."""

    def __init__(self, module: nn.Module, requires_grad: bool = True):
        self.requires_grad = bool(requires_grad)
        self.data: Optional[torch.Tensor] = None

        def capture(_module, _inputs, output):
            if self.requires_grad:
                output.requires_grad_(True).retain_grad()
            self.data = output

        self.hook = module.register_forward_hook(capture)

    def __enter__(self) -> "Hook":
        return self

    def __exit__(self, *exc) -> None:
        self.hook.remove()

    activation = property(lambda self: self.data, doc="the captured output tensor (None before the first forward)")
    gradient = property(lambda self: self.data.grad, doc="d loss / d activation after backward (requires_grad=True)")
