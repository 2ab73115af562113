"""detector_postprocess (reference detectron2/modeling/postprocessing.py:10-79), box fields only.
On the hot path the same scale/clip/drop-empty is fused into the detection gather kernel
(csrc/boxes.hip det_gather_kernel); this host version serves callers holding `Instances`."""
from ..structures import Instances


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        return results
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    return results[output_boxes.nonempty()]
