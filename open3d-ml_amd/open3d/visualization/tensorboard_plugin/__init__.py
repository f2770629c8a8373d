from . import summary   # noqa: F401
