from . import filters  # noqa: F401
