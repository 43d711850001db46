"""MI355X-native vectorised peg-in-hole environment (drop-in for the hot path of guodashun/peg-in-hole-gym)."""
from ._lib import PihError  # noqa: F401

__all__ = ["PihError"]
