"""Forward-hook helper with the reference's ``Hook`` contract (reference multimodal/attention_maps.py:83-105).

Only the hook used inside ``VisionEncoder.forward`` is on the hot path; the Grad-CAM plotting helpers of
the reference file are visualisation code and out of scope."""
from __future__ import annotations

import torch
import torch.nn as nn


class Hook:
    """Context manager that records a module's output (and, if asked, keeps its gradient)."""

    def __init__(self, module: nn.Module, requires_grad: bool = True):
        self.data = None
        self.requires_grad = requires_grad
        self.hook = module.register_forward_hook(self._record)

    def _record(self, module, inputs, output):
        self.data = output
        if self.requires_grad:
            output.requires_grad_(True)
            output.retain_grad()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self.hook.remove()

    @property
    def activation(self) -> torch.Tensor:
        return self.data

    @property
    def gradient(self) -> torch.Tensor:
        return self.data.grad
