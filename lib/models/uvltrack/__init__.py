from . import uvltrack  # noqa: F401
