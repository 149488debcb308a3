"""MI355X-native EdgeConv path for the DGCNN segmentation model of
BenCurran98/AHN3-dgcnn.pytorch (reference model.py), behind the reference's own Python
surface: knn, get_graph_feature, DGCNN.

The directory name is not a valid Python identifier; import it through the shim module
`ahn3_dgcnn_pytorch_amd` at the repository root.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401
from .model import DGCNN, get_graph_feature, knn  # noqa: F401
from . import blocks, optim  # noqa: F401,E402

__all__ = ["DGCNN", "get_graph_feature", "knn", "ops", "blocks", "optim"]
