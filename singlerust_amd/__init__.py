"""singlerust_amd — MI355X (gfx950) implementation of SingleRust's sparse count-matrix hot path.

Python here is the harness above the C ABI (include/srx.h): it mirrors the reference's
``memory::{processing, statistics, processing::dim_red}`` function names and argument order
(src/lib.rs:5-15) so tests read like the reference's own.  All arithmetic runs in
lib/libsrx_hip.so (hand-written HIP); nothing here computes on the CPU.
"""
from . import _ffi
from ._ffi import SrxError
from .anndata import Context, DeviceCsr, Direction, FeatureSelection, FlexValue, IMAnnData
from . import memory

__all__ = ["Context", "DeviceCsr", "Direction", "FeatureSelection", "FlexValue", "IMAnnData", "SrxError", "memory", "_ffi"]
