from .Logger import Logger  # noqa: F401
