"""`nms` / `batched_nms` with the reference's signatures (detectron2/layers/nms.py:6-29) on the gfx950
ballot kernels (bit-exact keep indices w.r.t. the CPU algorithm; see csrc/nms.hip)."""
from ..kernels import batched_nms, nms  # noqa: F401
