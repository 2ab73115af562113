"""SURVEY.md section 8(f) "next" rows: the 224x224 descriptor crops and the COCO json wire format."""
import pytest
import torch

from helpers import gold

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op", ["pad", "context"])
def test_get_crops_qe_matches_reference(op):
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.wire import get_crops_qe

    g = gold("crops")
    img = syn.synthetic_image(7, 300, 420)[None].to("cuda:0")
    insts = [Instances((300, 420), gt_boxes=Boxes(b[None].float())) for b in g["boxes"]]
    crops = get_crops_qe(img, insts, op).cpu()
    assert crops.shape == (len(insts), 3, 224, 224)
    assert torch.equal(crops[:, :, ::7, ::7], g["crops_" + op])                 # pure gather: bit-exact
    assert torch.allclose(crops.double().sum(dim=(1, 2, 3)), g["sum_" + op], rtol=1e-12, atol=0)
