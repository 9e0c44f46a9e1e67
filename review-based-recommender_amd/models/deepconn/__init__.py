from .deepconn import DeepCoNNpp  # noqa: F401
