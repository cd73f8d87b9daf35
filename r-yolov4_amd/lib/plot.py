"""Drop-in for the geometry of the reference's lib/plot.py on the device.

    rescale_boxes(boxes, current_dim, original_shape)            lib/plot.py:9-31, same arguments, in place on columns 0-3
    detections_to_polys(detections, img_size, shapes)            the per-image loop of Detect.save_results / plot_boxes
                                                                  (detect.py:32-37, lib/plot.py:48-51) as ONE launch for the batch
The drawing itself (cv.drawContours / putText / imwrite, lib/plot.py:55-70) is host-side cv2 code and is not rebuilt here.
"""
import torch

from .. import hip


def _run(dets, img_of_det, shapes, current_dim, rescale):
    n = dets.shape[0]
    polys = torch.empty((n, 4, 2), dtype=torch.float32, device=dets.device)
    hip.call("ryolo_dets_to_polys", hip.ptr(dets), None if img_of_det is None else hip.ptr(img_of_det),
             None if shapes is None else hip.ptr(shapes), int(current_dim), int(rescale), n, hip.ptr(polys), hip.stream())
    return polys


def rescale_boxes(boxes, current_dim, original_shape):
    """lib/plot.py:9-31: undo pad-to-square + resize on boxes[:, :4] (centre/size kept), IN PLACE like the reference; returns boxes."""
    hip.require_device(boxes, "rescale_boxes")
    if boxes.dtype != torch.float32 or not boxes.is_contiguous() or boxes.dim() != 2 or boxes.shape[1] != 7:
        raise RuntimeError("rescale_boxes: expected contiguous float32 detections [n, 7] as post_process returns them")
    shapes = torch.tensor([[int(original_shape[0]), int(original_shape[1])]], dtype=torch.int32, device=boxes.device)
    _run(boxes, None, shapes, current_dim, 1)
    return boxes


def detections_to_polys(detections, img_size, shapes):
    """detections: list of B tensors [n_i, 7] (post_process output); shapes: B original (h, w) pairs.  Returns (boxes [N, 7] rescaled
    to the original images, polys [N, 4, 2], counts list) — what plot_boxes computes per image before drawing."""
    counts = [int(d.shape[0]) for d in detections]
    dev = detections[0].device if detections else torch.device("cuda")
    for d in detections:
        hip.require_device(d, "detections_to_polys")
    dets = torch.cat([d.float().reshape(-1, 7) for d in detections], 0).contiguous() if detections else torch.zeros((0, 7), device=dev)
    img = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts, dtype=torch.int64)).to(dev)
    sh = torch.tensor([[int(h), int(w)] for h, w in shapes], dtype=torch.int32, device=dev).reshape(-1, 2)
    polys = _run(dets, img.contiguous(), sh, img_size, 1)
    return dets, polys, counts
