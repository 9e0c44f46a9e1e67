"""models/narre/layers.py exposes MyConv1d / NgramFeat (lines 119-153, 365-401 of the reference file are the
only classes narre.py imports; the other ~35 classes of that file are unreachable from any trainer,
SURVEY.md §2a row 4).  They are the same HIP-backed classes as models/deepconn/layers.py."""
from ..deepconn.layers import HierPooling, MyConv1d, NgramFeat  # noqa: F401
