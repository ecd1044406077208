"""``unipercept.nn.layers.deform_attn`` — the module path the reference's callers import.

Copy (or symlink) this directory to ``sources/unipercept/nn/layers/deform_attn`` of a unipercept
checkout; it forwards to the MI355X implementation in ``unipercept_amd``.
"""

from unipercept_amd.deform_attn import (  # noqa: F401
    MSDeformAttn,
    MultiScaleDeformAttnFunction,
    MultiScaleFlashAttnFunction,
    ms_deform_attn,
    reference,
)

__all__ = [
    "MSDeformAttn",
    "MultiScaleDeformAttnFunction",
    "MultiScaleFlashAttnFunction",
    "ms_deform_attn",
    "reference",
]
