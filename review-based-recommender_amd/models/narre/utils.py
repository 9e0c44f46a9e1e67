from ..deepconn.utils import masked_tensor  # noqa: F401
