"""omnibiote_amd — MI355X (gfx950) implementation of the OmniBioTE encoder-training hot path.

The compute lives in ``libomnibiote_hip.so`` (hand-written HIP, C ABI declared in ``include/omnibiote_hip.h``);
this package is the host side: a ctypes binding (``_lib``), tensor-level wrappers (``ops``), the drop-in
``model`` module (same class surface as the reference's ``training/model.py``) and the data-parallel training
harness (``train_encoder``).  There is no CPU fallback: every compute entry point raises if the HIP library is
missing or the tensors are not on a GPU.
"""
__version__ = "0.1.0"
