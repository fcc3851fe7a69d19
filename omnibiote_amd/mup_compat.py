"""Stand-ins for the three ``mup==1.0.0`` entry points the reference uses (README.md:16; model.py:19,208;
train_encoder.py:7,157-166,199).  The package is not vendored in the reference tree and not installable here, so
these restate its *published* behaviour for exactly the call pattern of the reference — **parity unpinned**
(SURVEY.md §8c): nothing in the reference pins these numbers, and they must be checked against the real package
on a machine that has it before claiming optimizer-trajectory parity.

  MuReadout(in, out, bias=False)        y = Linear(output_mult * x / width_mult),  width_mult = fan_in / base fan_in
  set_base_shapes(model, base, delta=)  tags every parameter with .infshape; rescales the readout weight by
                                        sqrt(width_mult) (once)
  MuAdamW(params, lr, ...)              AdamW with two kinds of groups: matrix-like parameters (two width-scaled
                                        dims) get lr / width_mult and weight_decay * width_mult; the rest are unchanged
If the real ``mup`` is importable it is used instead (see model.py / train_encoder.py).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Optional

import torch
import torch.nn as nn


class InfDim:
    def __init__(self, base_dim: Optional[int], dim: int):
        self.base_dim, self.dim = base_dim, dim

    def isinf(self) -> bool:
        return self.base_dim is not None

    def width_mult(self) -> float:
        return self.dim / self.base_dim if self.base_dim is not None else 1.0


class InfShape(tuple):
    def __new__(cls, dims):
        return super().__new__(cls, dims)

    def ninf(self) -> int:
        return sum(1 for d in self if d.isinf())

    def width_mult(self) -> float:
        # fan-in multiplier: the last dimension for matrices, the only one for vectors
        if len(self) == 0:
            return 1.0
        return self[-1].width_mult()


class MuReadout(nn.Linear):
    def __init__(self, *args, readout_zero_init=False, output_mult=1.0, **kwargs):
        self.output_mult = output_mult
        self.readout_zero_init = readout_zero_init
        super().__init__(*args, **kwargs)

    def reset_parameters(self) -> None:
        if getattr(self, "readout_zero_init", False):
            self.weight.data[:] = 0
            if self.bias is not None:
                self.bias.data[:] = 0
        else:
            super().reset_parameters()

    def width_mult(self) -> float:
        assert hasattr(self.weight, "infshape"), (
            "Please call set_base_shapes(...). If using torch.nn.DataParallel, switch to distributed training with "
            "torch.nn.parallel.DistributedDataParallel instead")
        return self.weight.infshape.width_mult()

    def _rescale_parameters(self) -> None:
        if hasattr(self, "_has_rescaled_params") and self._has_rescaled_params:
            raise RuntimeError("`_rescale_parameters` has been called once before already.")
        if self.bias is not None:
            self.bias.data *= self.width_mult() ** 0.5
        self.weight.data *= self.width_mult() ** 0.5
        self._has_rescaled_params = True

    def forward(self, x):
        return super().forward(self.output_mult * x / self.width_mult())


def set_base_shapes(model: nn.Module, base: nn.Module, rescale_params: bool = True, delta: Optional[nn.Module] = None):
    """Tag parameters with the dimensions that scale with width.  A dimension is 'infinite' when base and delta
    disagree on it (or, without delta, when base and model disagree)."""
    base_p = dict(base.named_parameters())
    delta_p = dict(delta.named_parameters()) if delta is not None else None
    for name, p in model.named_parameters():
        bs = base_p[name].shape
        ds = delta_p[name].shape if delta_p is not None else p.shape
        dims = []
        for b_dim, d_dim, dim in zip(bs, ds, p.shape):
            dims.append(InfDim(b_dim if b_dim != d_dim else None, dim))
        p.infshape = InfShape(dims)
    if rescale_params:
        for m in model.modules():
            if isinstance(m, MuReadout):
                m._rescale_parameters()
    return model


def mu_param_groups(params, lr: float, weight_decay: float):
    """Parameter groups of MuAdam(W): one per distinct width multiplier of the matrix-like parameters (two
    width-scaled dims), then one for everything else."""
    matrix_like = defaultdict(list)
    vector_like = []
    for p in params:
        assert hasattr(p, "infshape"), "A parameter has no infshape; call set_base_shapes on the model first"
        n = p.infshape.ninf()
        if n == 2:
            matrix_like[p.infshape.width_mult()].append(p)
        elif n > 2:
            raise NotImplementedError("more than 2 inf dimensions")
        else:
            vector_like.append(p)
    groups = []
    for wm, ps in matrix_like.items():
        groups.append({"params": ps, "lr": lr / wm, "weight_decay": weight_decay * wm})
    groups.append({"params": vector_like, "lr": lr, "weight_decay": weight_decay})
    return groups


def MuAdamW(params, lr=1e-3, weight_decay=1e-2, **kwargs):
    params = list(params)
    return torch.optim.AdamW(mu_param_groups(params, lr, weight_decay), lr=lr, weight_decay=weight_decay, **kwargs)
