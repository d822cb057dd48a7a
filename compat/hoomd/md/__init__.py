"""hoomd.md: mode_standard(dt) and the placeholder _md module."""
from . import _md, integrate   # noqa: F401
