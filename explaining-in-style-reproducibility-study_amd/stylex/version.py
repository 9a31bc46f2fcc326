"""Checkpoint-format version written into every ``model_<n>.pt`` and printed on load (same value as the
reference release this package is a drop-in for)."""

VERSION_TUPLE = (1, 8, 7)
__version__ = ".".join(str(v) for v in VERSION_TUPLE)
