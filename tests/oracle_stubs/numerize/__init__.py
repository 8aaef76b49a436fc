from . import numerize  # noqa: F401
