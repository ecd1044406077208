"""MI355X-native pieces of ``unipercept.vision``: PointRend point sampling."""

from . import pointrend

__all__ = ["pointrend"]
