"""lvc_amd: an MI355X-native (gfx950) implementation of the data-parallel hot path of prannaykaul/lvc --
the Faster-R-CNN-FPN `GeneralizedRCNN` forward and the label-verification kNN sweep -- behind the
reference's own registry / config / state_dict surface.  See DESIGN.md and include/lvc_amd.h."""
__version__ = "0.1.0"

from . import ops  # noqa: E402,F401  (registers torch.ops.lvc_amd.*; the native library itself loads on first call)
