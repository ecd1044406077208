"""Drop-in for ``unipercept.nn.layers.deform_attn`` on MI355X.

Exports what the reference's test imports from that package
(tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:6-10): ``MultiScaleDeformAttnFunction``,
``MultiScaleFlashAttnFunction`` and ``reference``; plus the functional alias ``ms_deform_attn``, the
``MSDeformAttn`` layer and ``SamplingPointsFunction`` (the layer's fused point epilogue).
"""

from . import reference
from ._function import (
    MultiScaleDeformAttnFunction,
    MultiScaleFlashAttnFunction,
    SamplingPointsFunction,
    ms_deform_attn,
)
from .encoder import (
    MSDeformAttnEncoder,
    MSDeformAttnEncoderLayer,
    encoder_reference_points,
    flatten_levels,
    sine_position_embedding,
)
from .layer import MSDeformAttn

__all__ = [
    "MSDeformAttn",
    "MSDeformAttnEncoder",
    "MSDeformAttnEncoderLayer",
    "encoder_reference_points",
    "flatten_levels",
    "sine_position_embedding",
    "MultiScaleDeformAttnFunction",
    "MultiScaleFlashAttnFunction",
    "SamplingPointsFunction",
    "ms_deform_attn",
    "reference",
]
