"""Import-time stand-in for `flatbuffers` (pyprob/ppx/*.py). PPX is out of scope."""
from . import compat, table  # noqa: F401


class Builder:
    def __init__(self, *a, **k):
        raise RuntimeError('flatbuffers stub')
