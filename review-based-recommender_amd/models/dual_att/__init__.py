from .dual_att import DualAtt  # noqa: F401
