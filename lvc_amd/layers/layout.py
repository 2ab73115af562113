"""Layout plumbing between the NCHW-shaped tensors of the module boundary and the NHWC storage the
gfx950 kernels use.  A feature map is kept as a contiguous [N,H,W,C] buffer; modules hand out its
`permute(0,3,1,2)` view (= torch channels_last), so shapes seen by callers are the reference's NCHW
while no byte is moved."""
import torch


def to_nhwc(x):
    """logical NCHW tensor -> contiguous [N,H,W,C] tensor (zero-copy when x is channels-last)."""
    assert x.dim() == 4
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def to_nchw_view(y):
    """contiguous [N,H,W,C] -> NCHW-shaped view."""
    return y.permute(0, 3, 1, 2)


def require_device(x, what):
    if not x.is_cuda:
        raise RuntimeError(
            "{}: lvc_amd runs on MI355X only (tensor is on {}); set MODEL.DEVICE to 'cuda'. "
            "There is no CPU fallback in the product path.".format(what, x.device))
