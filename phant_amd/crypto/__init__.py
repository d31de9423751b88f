from . import hasher  # noqa: F401
