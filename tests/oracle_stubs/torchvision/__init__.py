from . import utils, transforms  # noqa: F401
