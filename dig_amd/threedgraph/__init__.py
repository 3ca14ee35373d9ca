from . import evaluation, method, utils  # noqa: F401
