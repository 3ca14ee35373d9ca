from . import dataset, evaluation, method, utils  # noqa: F401
