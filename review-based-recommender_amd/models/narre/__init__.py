from .narre import NARRE  # noqa: F401
