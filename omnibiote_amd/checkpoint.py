"""Checkpoint compatibility with the reference.

The reference checkpoints whole objects: ``torch.save(model.module, f"{save_name}_{tokens}.pt")``
(training/train_encoder.py:413,430), and its eval scripts ``torch.load`` them, call ``.to(bfloat16)`` and fine-tune
(evals/gue.py:279).  Such a pickle names its classes by module path — ``model.OmniBioTA``, ``model.Block`` ...,
``mup.layer.MuReadout`` — so loading one normally imports the reference's ``model`` module.

``load_checkpoint`` unpickles those files into THIS package's classes instead (same attribute tree and state_dict, HIP
kernels underneath) by redirecting the class lookups; files written by ``save_checkpoint`` (whole-object pickle of this
package's classes, the same convention) load through it as well.  Tensors, buffers — including a ``freqs_cis`` already
degraded to real bf16 by the reference's ``.to(bfloat16)`` — and per-parameter ``infshape`` attributes come through as
pickled.
"""
from __future__ import annotations

import pickle
from typing import Any

import torch

_REDIRECT = {
    ("model", "OmniBioTA"), ("model", "OmniBioTAConfig"), ("model", "Block"), ("model", "SelfAttention"), ("model", "MLP"),
    ("model", "LayerNorm"),
}
_READOUT = {("mup.layer", "MuReadout"), ("mup", "MuReadout")}
_INFSHAPE = {("mup.infshape", "InfShape"): "InfShape", ("mup.infshape", "InfDim"): "InfDim"}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        from . import model as M
        from . import mup_compat
        if (module, name) in _REDIRECT:
            return getattr(M, name)
        if (module, name) in _READOUT:
            return M.MuReadout
        if (module, name) in _INFSHAPE:
            try:
                return super().find_class(module, name)      # the real mup, when installed
            except Exception:
                return getattr(mup_compat, _INFSHAPE[(module, name)])
        return super().find_class(module, name)


class _PickleModule:
    """The minimal ``pickle_module`` interface torch.load needs."""
    __name__ = "omnibiote_amd.checkpoint"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler


def load_checkpoint(path: str, map_location="cpu"):
    """Load a whole-object model checkpoint written by the reference trainer or by ``save_checkpoint``."""
    obj = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    for m in getattr(obj, "modules", lambda: [])():
        # reference SelfAttention objects carry no RoPE cache slots; give every module the attributes this package expects
        if m.__class__.__name__ == "SelfAttention":
            m.__dict__.setdefault("_rope_key", None)
            m.__dict__.setdefault("_rope_val", None)
        if m.__class__.__name__ == "MuReadout":
            m.__dict__.setdefault("output_mult", 1.0)
            m.__dict__.setdefault("readout_zero_init", False)
    return obj


def save_checkpoint(model, path: str) -> None:
    """Whole-object pickle, the reference's convention (train_encoder.py:413).  ``model`` may be DDP-wrapped."""
    torch.save(model.module if hasattr(model, "module") else model, path)
