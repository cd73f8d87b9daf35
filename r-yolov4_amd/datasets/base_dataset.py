"""Device-side end of the reference's datasets/base_dataset.py (SURVEY §8(f) N2 slice): everything BaseDataset.__getitem__ does
AFTER the cv2 stages (mosaic / warp / hsv) plus collate_fn, for a whole batch in three launches.

    finalize_batch(imgs_u8, targets10, flags, csl)     base_dataset.py:129-157 + collate_fn :159-166
    gaussian_label(label, num_class, u, sig)           base_dataset.py:13-31 (host numpy, as in the reference)

`imgs_u8` [B, S, S, 3] uint8 BGR and `targets10` [nt, 10] = (image slot, class, x1, y1, ..., x4, y4 in pixels) are what the
reference holds at line 128; `flags[b]` bit 0 / 1 = the fliplr / flipud decisions (hyp['fliplr'], hyp['flipud'] draws, :133-138).
Returns (imgs [B, 3, S, S] fp32 RGB in [0, 1], targets [n, 7 | 187]) exactly as collate_fn hands them to train.py:183.
The cv2 stages themselves (imread, resize, hsv, mosaic, warpPerspective, mixup) are not rebuilt (no cv2 in this image to pin them).
"""
import numpy as np
import torch

from .. import hip


def gaussian_label(label, num_class, u=0, sig=4.0):
    """Circular smooth label of one angle class (host numpy, same signature as the reference's): the gaussian window over the
    class axis, rotated so that its peak sits at `label`; `int()` truncates toward zero exactly like the reference's index."""
    half = num_class / 2
    window = np.exp(-np.square(np.arange(-half, half) - u) / (2.0 * sig * sig))
    return np.roll(window, -int(half - label))


def finalize_batch(imgs_u8, targets10, flags=None, csl=False):
    hip.require_device(imgs_u8, "finalize_batch")
    if imgs_u8.dtype != torch.uint8 or imgs_u8.dim() != 4 or imgs_u8.shape[3] != 3 or not imgs_u8.is_contiguous():
        raise RuntimeError("finalize_batch: images must be a contiguous uint8 tensor [B, H, W, 3] (BGR)")
    dev = imgs_u8.device
    B, H, W, _ = imgs_u8.shape
    if flags is not None:
        flags = flags.to(device=dev, dtype=torch.uint8).contiguous()
        if flags.numel() != B:
            raise RuntimeError("finalize_batch: one flip flag byte per image")
    imgs = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    hip.call("ryolo_to_tensor", hip.ptr(imgs_u8), B, H, W, None if flags is None else hip.ptr(flags), hip.ptr(imgs), hip.stream())
    tg = targets10.to(device=dev, dtype=torch.float32).contiguous()
    if tg.dim() != 2 or (tg.shape[0] and tg.shape[1] != 10):
        raise RuntimeError("finalize_batch: targets must be [nt, 10] = (image slot, class, 8 polygon coordinates)")
    nt = tg.shape[0]
    ncols = 187 if csl else 7
    out = torch.empty((nt, ncols), dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(nt, 1), dtype=torch.int32, device=dev) if csl else None
    hip.call("ryolo_encode_labels", hip.ptr(tg) if nt else None, nt, H, W, None if flags is None else hip.ptr(flags), None, 1 if csl else 0,
             hip.ptr(out) if nt else None, count.data_ptr(), None if ws is None else ws.data_ptr(), hip.stream())
    n = int(count.item()) if nt else 0            # the one host read: collate_fn's torch.cat needs the row count too
    return imgs, out[:n]
