"""``unipercept.vision.pointrend`` — the sampling functions of the module the reference's callers
import, forwarded to the MI355X implementation in ``unipercept_amd``.

In a unipercept checkout, replace the bodies of ``sample`` (and, if wanted, the three point
generators) in ``sources/unipercept/vision/pointrend.py`` by these imports.
"""

from unipercept_amd.vision.pointrend import (  # noqa: F401
    BinsMode,
    bins_by_values,
    logit_uncertainty,
    map_in_bins,
    random_points,
    random_points_with_bins,
    random_points_with_importance,
    sample,
)

__all__ = [
    "BinsMode",
    "bins_by_values",
    "logit_uncertainty",
    "map_in_bins",
    "random_points",
    "random_points_with_bins",
    "random_points_with_importance",
    "sample",
]
