"""Device-side mosaic / warp / hsv / mixup of the reference's loader (datasets/base_dataset.py:83-128,170-330,
lib/augmentations.py:8-74) over a uint8 image cache resident in HBM.

Split of the work (as in the reference, but per batch instead of per sample and per worker process):
  host     the random draws and the rectangle / label arithmetic — a few integers and a few dozen polygons per sample, kept in the
           reference's exact order of operations (mosaic4_plan / mosaic9_plan / load_target / warp_matrix / warp_targets);
  device   every pixel: paste (csrc/augment.hip paste_rects), perspective warp, hsv gain, mixup — then finalize_batch
           (datasets/base_dataset.py of this package) turns the uint8 canvases into the fp32 training batch.
Parity: plans, labels, paste and mixup are pinned to the imported reference (tests/golden/make_golden_aug.py ran the real
load_mosaic / load_mosaic9 / load_target / mixup / random_warping label code); the pixels of warp and hsv restate OpenCV (absent
here, version un-pinned by the reference): parity unpinned, checked against oracle/ref_data.py's numpy restatement only.
"""
import ctypes
import math

import numpy as np
import torch

from .. import hip


class ImagePool:
    """uint8 HWC (BGR) images of different sizes packed into one device buffer (what cv2.imread + resize + hsv leave per file)."""

    def __init__(self, images, device):
        self.shapes = [tuple(im.shape[:2]) for im in images]
        sizes = [h * w * 3 for h, w in self.shapes]
        self.offsets = [0]
        for s in sizes[:-1]:
            self.offsets.append(self.offsets[-1] + ((s + 15) // 16) * 16)
        total = self.offsets[-1] + sizes[-1] if sizes else 0
        host = torch.zeros(max(total, 1), dtype=torch.uint8)
        for im, o, s in zip(images, self.offsets, sizes):
            host[o:o + s] = torch.as_tensor(np.ascontiguousarray(im)).reshape(-1)
        self.buf = host.to(device)


# ------------------------------------------------------------------------------------------------ plans (host integers)
def mosaic4_plan(s, shapes, yc, xc):
    """datasets/base_dataset.py:224-268: for the 4 (h, w) shapes -> [(x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b)] on the 2s x 2s canvas."""
    out = []
    for i, (h, w) in enumerate(shapes):
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        out.append((x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b))
    return out


def mosaic9_plan(s, shapes):
    """datasets/base_dataset.py:270-315: placement of the 9 images on the 3s x 3s canvas -> [(padx, pady, x1, y1, x2, y2)]."""
    out = []
    hp = wp = h_ = w_ = 0
    for i, (h, w) in enumerate(shapes):
        if i == 0:
            h_, w_ = h, w
            c = s, s, s + w, s + h
        elif i == 1:
            c = s, s - h, s + w, s
        elif i == 2:
            c = s + wp, s - h, s + wp + w, s
        elif i == 3:
            c = s + w_, s, s + w_ + w, s + h
        elif i == 4:
            c = s + w_, s + hp, s + w_ + w, s + hp + h
        elif i == 5:
            c = s + w_ - w, s + h_, s + w_, s + h_ + h
        elif i == 6:
            c = s + w_ - wp - w, s + h_, s + w_ - wp, s + h_ + h
        elif i == 7:
            c = s - w, s + h_ - h, s, s + h_
        else:
            c = s - w, s + h_ - hp - h, s, s + h_ - hp
        padx, pady = c[:2]
        x1, y1, x2, y2 = [max(x, 0) for x in c]
        out.append((padx, pady, x1, y1, x2, y2))
        hp, wp = h, w
    return out


def filtering(targets, boarder):
    """datasets/base_dataset.py:332-345 (strict inequalities on the mean vertex)."""
    x1, x2, y1, y2 = boarder
    x = torch.mean(targets[:, [2, 4, 6, 8]], dim=1)
    y = torch.mean(targets[:, [3, 5, 7, 9]], dim=1)
    return targets[(x > x1) & (x < x2) & (y > y1) & (y < y2)]


def load_target(polys, labels, pad, img_size0, img_size, normalized_labels, boarder=None):
    """datasets/base_dataset.py:188-222 after the label file was parsed: polys [n, 8] float32 (modified in place like the
    reference's), labels [n] -> targets [m, 10] in canvas pixels."""
    if not len(labels):
        return torch.zeros((0, 10))
    if not normalized_labels:
        h0, w0 = img_size0
        polys[:, [0, 2, 4, 6]] /= w0
        polys[:, [1, 3, 5, 7]] /= h0
    h_, w_ = img_size
    polys[:, [0, 2, 4, 6]] *= w_
    polys[:, [1, 3, 5, 7]] *= h_
    targets = torch.zeros((len(labels), 10))
    targets[:, 1:] = torch.cat((labels.unsqueeze(-1), polys), -1)
    if boarder is not None:
        targets = filtering(targets, boarder)
    targets[:, [2, 4, 6, 8]] += pad[1]
    targets[:, [3, 5, 7, 9]] += pad[0]
    return targets


def warp_matrix(shape, a, s, tx, ty, border=(0, 0)):
    """lib/augmentations.py:45-65 with the four random draws made explicit: rotation angle a (deg), scale s, translation
    fractions tx, ty.  Returns (M 3x3 float64, (width, height)).  cv2.getRotationMatrix2D is OpenCV's documented closed form."""
    height = shape[0] + border[0] * 2
    width = shape[1] + border[1] * 2
    C = np.eye(3)
    C[0, 2] = -shape[1] / 2
    C[1, 2] = -shape[0] / 2
    R = np.eye(3)
    ang = a * math.pi / 180.0
    alpha, beta = s * math.cos(ang), s * math.sin(ang)
    R[:2] = np.array([[alpha, beta, 0.0], [-beta, alpha, 0.0]])      # center (0, 0): both translation terms vanish
    T = np.eye(3)
    T[0, 2] = tx * width
    T[1, 2] = ty * height
    return T @ R @ C, (width, height)


def warp_targets(targets, M):
    """lib/augmentations.py:67-74: polygon vertices through M in double, written back into the float32 targets (in place)."""
    Mt = torch.tensor(M, dtype=torch.double)
    pts = targets[:, 2:].reshape(-1, 2)
    pts = torch.cat((pts, torch.ones(pts.size()[0]).view(pts.size()[0], 1)), dim=-1).double()
    pts = (torch.matmul(Mt, pts.t())).t()[:, :2]
    targets[:, 2:] = pts.reshape(-1, 8)
    return targets


# ------------------------------------------------------------------------------------------------ pixels (device)
class _Rect(ctypes.Structure):
    _fields_ = [("src_off", ctypes.c_int64), ("src_w", ctypes.c_int), ("sx", ctypes.c_int), ("sy", ctypes.c_int), ("dx", ctypes.c_int),
                ("dy", ctypes.c_int), ("w", ctypes.c_int), ("h", ctypes.c_int), ("canvas", ctypes.c_int)]


def paste(pool, rects, ncanvas, CH, CW, fill=114):
    """rects: [(image index in the pool, sx, sy, dx, dy, w, h, canvas index)] in paste order -> canvases [ncanvas, CH, CW, 3] uint8."""
    n = ctypes.c_int()
    hip.call("ryolo_paste_rect_bytes", n)
    assert n.value == ctypes.sizeof(_Rect)
    arr = (_Rect * max(len(rects), 1))()
    k = 0
    for (img, sx, sy, dx, dy, w, h, cv) in rects:
        if w <= 0 or h <= 0:
            continue                                              # empty numpy slices paste nothing
        arr[k] = _Rect(pool.offsets[img], pool.shapes[img][1], sx, sy, dx, dy, w, h, cv)
        k += 1
    dev = pool.buf.device
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    canvas = torch.empty((ncanvas, CH, CW, 3), dtype=torch.uint8, device=dev)
    hip.call("ryolo_paste_rects", hip.ptr(pool.buf), hip.ptr(table), k, hip.ptr(canvas), ncanvas, CH, CW, fill, hip.stream())
    return canvas


def mosaic4(pool, indices, s, yc, xc, canvas=0):
    """Rectangles of one 4-image mosaic (datasets/base_dataset.py:224-268) for paste(): returns (rects, [(padh, padw), boarder] per image)."""
    rects, meta = [], []
    for img, (x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b) in zip(indices, mosaic4_plan(s, [pool.shapes[i] for i in indices], yc, xc)):
        # numpy slice assignment img4[y1a:y2a, x1a:x2a] = img[y1b:y2b, x1b:x2b]: both sides have the same extent by construction
        rects.append((img, x1b, y1b, x1a, y1a, x2a - x1a, y2a - y1a, canvas))
        meta.append(((y1a - y1b, x1a - x1b), (x1b, x2b, y1b, y2b)))
    return rects, meta


def mosaic9(pool, indices, s, yc, xc, canvas=0):
    """One 9-image mosaic cropped to [yc, yc + 2s) x [xc, xc + 2s) (datasets/base_dataset.py:270-330): the crop is folded into the
    destination coordinates.  Returns (rects, [(pady, padx), boarder] per image)."""
    rects, meta = [], []
    for img, (padx, pady, x1, y1, x2, y2) in zip(indices, mosaic9_plan(s, [pool.shapes[i] for i in indices])):
        h, w = pool.shapes[img]
        # img9[y1:y2, x1:x2] = img[y1 - pady:, x1 - padx:]  (canvas 3s x 3s clips x2 / y2; the source slice runs to the image end)
        sx, sy = x1 - padx, y1 - pady
        ww, hh = min(x2, 3 * s) - x1, min(y2, 3 * s) - y1
        ww, hh = min(ww, w - sx), min(hh, h - sy)
        dx, dy = x1 - xc, y1 - yc
        # clip against the crop window
        cx0, cy0 = max(0, -dx), max(0, -dy)
        cw, ch = min(ww, 2 * s - dx) - cx0, min(hh, 2 * s - dy) - cy0
        rects.append((img, sx + cx0, sy + cy0, dx + cx0, dy + cy0, cw, ch, canvas))
        meta.append(((pady, padx), (x1 - padx, w, y1 - pady, h)))
    return rects, meta


def warp_perspective(imgs, Ms, dsize, border=114):
    """imgs [B, H, W, 3] uint8 on the device, Ms [B] 3x3 (what the reference hands to cv2.warpPerspective), dsize (width, height)."""
    hip.require_device(imgs, "warp_perspective")
    B, H, W, _ = imgs.shape
    DW, DH = dsize
    minv = torch.tensor(np.stack([np.linalg.inv(np.asarray(M, dtype=np.float64)) for M in Ms]).reshape(B, 9), dtype=torch.float64).to(imgs.device)
    out = torch.empty((B, DH, DW, 3), dtype=torch.uint8, device=imgs.device)
    hip.call("ryolo_warp_perspective_u8", hip.ptr(imgs.contiguous()), B, H, W, hip.ptr(minv), hip.ptr(out), DH, DW, border, hip.stream())
    return out


def hsv_luts(r):
    """lib/augmentations.py:13-17: the three uint8 LUTs for gains r = (hue, sat, val)."""
    x = np.arange(0, 256, dtype=np.asarray(r).dtype)
    return np.stack((((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8)))


def hsv_gain(img, r):
    """In place on a device uint8 image [..., 3] (BGR): lib/augmentations.py:8-21 with the random gains r given."""
    hip.require_device(img, "hsv_gain")
    lut = torch.from_numpy(hsv_luts(np.asarray(r, dtype=np.float64))).to(img.device).contiguous()
    hip.call("ryolo_hsv_gain_u8", hip.ptr(img), img.numel() // 3, hip.ptr(lut), hip.stream())
    return img


def mixup(img, img2, r):
    """lib/augmentations.py:24-28 (image part): uint8(img * r + img2 * (1 - r))."""
    hip.require_device(img, "mixup")
    out = torch.empty_like(img)
    hip.call("ryolo_mixup_u8", hip.ptr(img.contiguous()), hip.ptr(img2.contiguous()), float(r), img.numel(), hip.ptr(out), hip.stream())
    return out


def pad_to_square_plan(shape, new_shape):
    """datasets/base_dataset.py:33-56, the integer part: (new_unpad (w, h), (top, bottom, left, right), (dh, dw))."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, (top, bottom, left, right), (dh, dw)


def pad_to_square(img, new_shape, pad_value=114):
    """Device letterbox: img [H, W, 3] uint8 -> (canvas [top + NH + bottom, left + NW + right, 3], (dh, dw)) as the reference returns."""
    hip.require_device(img, "pad_to_square")
    H, W, _ = img.shape
    (nw, nh), (top, bottom, left, right), pad = pad_to_square_plan((H, W), new_shape)
    OH, OW = top + nh + bottom, left + nw + right
    out = torch.empty((OH, OW, 3), dtype=torch.uint8, device=img.device)
    hip.call("ryolo_letterbox_u8", hip.ptr(img.contiguous()), H, W, nh, nw, top, left, hip.ptr(out), OH, OW, pad_value, hip.stream())
    return out, pad
