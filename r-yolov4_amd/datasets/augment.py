"""Device-side sample composition of the reference's loader (datasets/base_dataset.py:83-128,170-330,
lib/augmentations.py:8-74) over a uint8 image cache resident in HBM.

The reference builds every training sample on a CPU worker with cv2.  Here nothing per pixel and nothing per label coordinate happens on
the host: the host only draws the random numbers and turns them into small TABLES — which rectangle of which cached image lands where
(placements), which 3x3 matrix a canvas is warped with, which parameters a label row is carried through — and the device applies them to
the whole batch, one launch per stage (csrc/augment.hip):

    resize_hsv_batch   load_image's cv2.resize + hsv for every source image a batch uses       (base_dataset.py:170-186)
    paste_rects        mosaic-4 / mosaic-9 assembly (and the letterbox paste) from a rectangle table   (:224-330)
    warp_perspective   random_warping's cv2.warpPerspective, one matrix per canvas               (lib/augmentations.py:45-65)
    mixup              uint8(a r + b (1 - r))                                                    (lib/augmentations.py:24-28)
    label_stage        load_target / mosaic crop / vertex warp of every label row, element-wise   (:188-222,318-330; :67-74)

A mosaic placement is data, not code: image i of a 4-mosaic hangs on the mosaic centre by one of its corners (MOSAIC4_CORNER), image i
of a 9-mosaic sits at an origin that is a fixed integer combination of the sizes of the first, the previous and the current image
(MOSAIC9_ORIGIN); everything else — the clipped source rectangle, the label shift, the label filter window — follows from intersecting
the placed rectangle with a window (`place`).

Parity: placements, paste, mixup and labels are pinned to the imported reference (fixtures G11 / G13: the real load_mosaic /
load_mosaic9 / load_target / mixup / random_warping / __getitem__ ran); the pixels of resize, warp and hsv restate OpenCV (absent here,
version un-pinned by the reference): parity unpinned, checked against oracle/ref_data.py's numpy restatement only.
"""
import ctypes
import math
from collections import namedtuple

import numpy as np
import torch

from .. import hip

_I, _L, _P = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p


class ImagePool:
    """uint8 HWC (BGR) images of different sizes packed into one device buffer (what cv2.imread leaves per file)."""

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


# ------------------------------------------------------------------------------------------------ placements (host integers)
# 4-mosaic (base_dataset.py:224-268): quadrant i touches the centre (xc, yc) with the corner named here — (1, 1) = its bottom-right
# corner, i.e. origin = centre - (w, h); (0, 0) = its top-left corner, origin = centre.
MOSAIC4_CORNER = ((1, 1), (0, 1), (1, 0), (0, 0))
# 9-mosaic (base_dataset.py:270-315): origin of image i on the 3s canvas = (s, s) + (ax . (w0, wp, w), ay . (h0, hp, h)) with
# (w0, h0) the first, (wp, hp) the previous and (w, h) the current image: centre, top, top-right, right, bottom-right, bottom,
# bottom-left, left, top-left — each image abuts the previous one.
MOSAIC9_ORIGIN = (((0, 0, 0), (0, 0, 0)), ((0, 0, 0), (0, 0, -1)), ((0, 1, 0), (0, 0, -1)), ((1, 0, 0), (0, 0, 0)), ((1, 0, 0), (0, 1, 0)),
                  ((1, 0, -1), (1, 0, 0)), ((1, -1, -1), (1, 0, 0)), ((0, 0, -1), (1, 0, -1)), ((0, 0, -1), (1, -1, -1)))

Placed = namedtuple("Placed", "sx sy dx dy w h")        # canvas[dy:dy+h, dx:dx+w] = image[sy:sy+h, sx:sx+w]; w or h <= 0: nothing


def place(ox, oy, w, h, win):
    """Image of size (w, h) with its top-left corner at (ox, oy); win = (x0, y0, x1, y1): the part inside the window, in source
    coordinates and in coordinates relative to the window's origin."""
    x0, y0 = max(ox, win[0]), max(oy, win[1])
    x1, y1 = min(ox + w, win[2]), min(oy + h, win[3])
    return Placed(x0 - ox, y0 - oy, x0 - win[0], y0 - win[1], x1 - x0, y1 - y0)


def mosaic4_origins(shapes, yc, xc):
    return [(xc - qx * w, yc - qy * h) for (h, w), (qx, qy) in zip(shapes, MOSAIC4_CORNER)]


def mosaic9_origins(s, shapes):
    out = []
    (h0, w0), (hp, wp) = shapes[0], (0, 0)
    for (h, w), (ax, ay) in zip(shapes, MOSAIC9_ORIGIN):
        out.append((s + ax[0] * w0 + ax[1] * wp + ax[2] * w, s + ay[0] * h0 + ay[1] * hp + ay[2] * h))
        hp, wp = h, w
    return out


# one source image placed on a canvas: paste rectangle + what its label rows need (load_target's pad and `boarder`, the crop window)
Use = namedtuple("Use", "img rect pad border crop")


def mosaic4_uses(shapes, indices, s, yc, xc):
    """4 images around (xc, yc) on the 2s x 2s canvas.  Labels: shifted by the image origin, kept when their mean vertex lies strictly
    inside the visible source rectangle (base_dataset.py:262-264)."""
    uses = []
    for img, (h, w), (ox, oy) in zip(indices, shapes, mosaic4_origins(shapes, yc, xc)):
        r = place(ox, oy, w, h, (0, 0, 2 * s, 2 * s))
        uses.append(Use(img, r, (oy, ox), (r.sx, r.sx + r.w, r.sy, r.sy + r.h), None))
    return uses


def mosaic9_uses(shapes, indices, s, yc, xc):
    """9 images on the 3s x 3s canvas, of which the window [xc, xc + 2s) x [yc, yc + 2s) is kept (base_dataset.py:270-330): the crop is
    folded into the paste coordinates; labels are filtered against the source rectangle left of / above which the canvas clipped the
    image (:311), shifted by the origin, filtered against the crop window and shifted by its corner (:321-328)."""
    uses = []
    for img, (h, w), (ox, oy) in zip(indices, shapes, mosaic9_origins(s, shapes)):
        on_canvas = place(ox, oy, w, h, (0, 0, 3 * s, 3 * s))
        r = place(ox, oy, w, h, (max(xc, 0), max(yc, 0), min(xc + 2 * s, 3 * s), min(yc + 2 * s, 3 * s)))
        r = r._replace(dx=r.dx + max(xc, 0) - xc, dy=r.dy + max(yc, 0) - yc)
        uses.append(Use(img, r, (oy, ox), (on_canvas.sx, w, on_canvas.sy, h), (xc, xc + 2 * s, yc, yc + 2 * s)))
    return uses


def affine(tx=0.0, ty=0.0, a_deg=0.0, scale=1.0):
    """3x3 float64: rotation by a_deg about the origin (OpenCV's getRotationMatrix2D convention: positive = counter-clockwise in image
    coordinates, i.e. [[c, s], [-s, c]]) times `scale`, then translation by (tx, ty)."""
    m = np.eye(3)
    ang = a_deg * math.pi / 180.0
    m[0, 0] = m[1, 1] = scale * math.cos(ang)
    m[0, 1] = scale * math.sin(ang)
    m[1, 0] = -m[0, 1]
    m[0, 2], m[1, 2] = tx, ty
    return m


def warp_matrix(shape, a, s, tx, ty, border=(0, 0)):
    """Matrix of random_warping (lib/augmentations.py:45-65) for its four draws: rotation angle a (deg), scale s, translation fractions
    tx, ty.  The image centre goes to the origin, is rotated / scaled there, and the result is moved by (tx, ty) of the OUTPUT size
    (input size + 2 * border).  Returns (M 3x3 float64, (width, height) of the output)."""
    height, width = shape[0] + border[0] * 2, shape[1] + border[1] * 2
    M = affine(tx * width, ty * height) @ affine(a_deg=a, scale=s) @ affine(-shape[1] / 2, -shape[0] / 2)
    return M, (width, height)


# ------------------------------------------------------------------------------------------------ tables -> device
class _Rect(ctypes.Structure):
    _fields_ = [("src_off", _L), ("src_w", _I), ("sx", _I), ("sy", _I), ("dx", _I), ("dy", _I), ("w", _I), ("h", _I), ("canvas", _I)]


class _ResizeItem(ctypes.Structure):
    _fields_ = [("src_off", _L), ("dst_off", _L), ("SH", _I), ("SW", _I), ("NH", _I), ("NW", _I), ("interp", _I), ("lut", _I)]


class _LabelRow(ctypes.Structure):
    _fields_ = [("poly", ctypes.c_float * 8), ("cls", ctypes.c_float), ("slot", _I), ("w0", ctypes.c_float), ("h0", ctypes.c_float),
                ("w1", ctypes.c_float), ("h1", ctypes.c_float), ("bx1", ctypes.c_float), ("bx2", ctypes.c_float), ("by1", ctypes.c_float),
                ("by2", ctypes.c_float), ("padw", ctypes.c_float), ("padh", ctypes.c_float), ("cx1", ctypes.c_float), ("cx2", ctypes.c_float),
                ("cy1", ctypes.c_float), ("cy2", ctypes.c_float), ("mat", _I)]


LABEL_ROW_DTYPE = np.dtype([("poly", np.float32, 8), ("cls", np.float32), ("slot", np.int32), ("w0", np.float32), ("h0", np.float32),
                            ("w1", np.float32), ("h1", np.float32), ("bx1", np.float32), ("bx2", np.float32), ("by1", np.float32),
                            ("by2", np.float32), ("padw", np.float32), ("padh", np.float32), ("cx1", np.float32), ("cx2", np.float32),
                            ("cy1", np.float32), ("cy2", np.float32), ("mat", np.int32)])
INTERP_LINEAR, INTERP_AREA, INTERP_COPY = 0, 1, 2
_checked = False


def _check_layouts():
    global _checked
    if _checked:
        return
    n = _I()
    for name, t in (("ryolo_paste_rect_bytes", _Rect), ("ryolo_resize_item_bytes", _ResizeItem), ("ryolo_label_row_bytes", _LabelRow)):
        hip.call(name, n)
        if n.value != ctypes.sizeof(t):
            raise RuntimeError(f"ryolov4_amd: struct layout mismatch for {t.__name__}: C {n.value} vs ctypes {ctypes.sizeof(t)}")
    if LABEL_ROW_DTYPE.itemsize != ctypes.sizeof(_LabelRow):
        raise RuntimeError("ryolov4_amd: LABEL_ROW_DTYPE drifted from LabelRow")
    _checked = True


def _to_device(arr, dev):
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)


def paste(src_buf, rects, ncanvas, CH, CW, fill=114):
    """rects: [(source byte offset, source row pitch in pixels, Placed, canvas index)] in paste order (later rectangles win) ->
    canvases [ncanvas, CH, CW, 3] uint8, `fill` where nothing was pasted."""
    _check_layouts()
    live = [(off, pitch, r, cv) for off, pitch, r, cv in rects if r.w > 0 and r.h > 0]         # empty numpy slices paste nothing
    arr = (_Rect * max(len(live), 1))()
    for k, (off, pitch, r, cv) in enumerate(live):
        arr[k] = _Rect(off, pitch, r.sx, r.sy, r.dx, r.dy, r.w, r.h, cv)
    dev = src_buf.device
    table = _to_device(arr, dev)
    canvas = torch.empty((ncanvas, CH, CW, 3), dtype=torch.uint8, device=dev)
    hip.call("ryolo_paste_rects", hip.ptr(src_buf), hip.ptr(table), len(live), hip.ptr(canvas), ncanvas, CH, CW, fill, hip.stream())
    return canvas


def pool_rects(pool, uses, canvas=0):
    """paste() rows for source images read straight from an ImagePool (no resize / hsv stage in between)."""
    return [(pool.offsets[u.img], pool.shapes[u.img][1], u.rect, canvas) for u in uses]


def resize_hsv_batch(pool, items, luts=None):
    """items: [(pool image index, (NH, NW), interp, lut index or -1)] -> (staging buffer, [byte offset of every resized image]).
    luts: uint8 [n, 3, 256] (hsv_luts per image) or None."""
    _check_layouts()
    dev = pool.buf.device
    arr = (_ResizeItem * max(len(items), 1))()
    offs, total, maxpix = [], 0, 0
    for k, (img, (nh, nw), interp, lut) in enumerate(items):
        sh, sw = pool.shapes[img]
        arr[k] = _ResizeItem(pool.offsets[img], total, sh, sw, nh, nw, interp, lut)
        offs.append(total)
        total += ((nh * nw * 3 + 15) // 16) * 16
        maxpix = max(maxpix, nh * nw)
    stage = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
    if items:
        lt = None if luts is None or not len(luts) else torch.as_tensor(np.ascontiguousarray(luts, dtype=np.uint8)).to(dev)
        hip.call("ryolo_resize_hsv_batch", hip.ptr(pool.buf), hip.ptr(_to_device(arr, dev)), len(items), maxpix, hip.ptr(lt), hip.ptr(stage),
                 hip.stream())
    return stage, offs


def label_stage(rows, mats, device):
    """rows: numpy structured array (LABEL_ROW_DTYPE), mats: [n, 3, 3] float64 warp matrices (or empty) -> targets10 [nrows, 10] on the
    device = (slot, class, 8 vertex coordinates), NaN vertices for rows a filter dropped (ryolo_encode_labels removes them in order)."""
    _check_layouts()
    n = len(rows)
    out = torch.empty((n, 10), dtype=torch.float32, device=device)
    if n:
        table = torch.from_numpy(np.frombuffer(rows.tobytes(), dtype=np.uint8).copy()).to(device)
        mt = None if mats is None or not len(mats) else torch.as_tensor(np.ascontiguousarray(mats, dtype=np.float64).reshape(-1, 9)).to(device)
        hip.call("ryolo_label_stage", hip.ptr(table), n, hip.ptr(mt), hip.ptr(out), hip.stream())
    return out


def label_rows(polys, labels, slot, img_size0, img_size, use, mat=-1, normalized_labels=False):
    """The LABEL_ROW_DTYPE rows of one placed source image: its parsed polygons [n, 8] / classes [n] with the parameters of
    load_target (original and resized size, `boarder`, pad), of the mosaic-9 crop and the index of the canvas' warp matrix."""
    n = len(labels)
    rows = np.zeros(n, dtype=LABEL_ROW_DTYPE)
    if not n:
        return rows
    rows["poly"], rows["cls"], rows["slot"], rows["mat"] = polys, labels, slot, mat
    if not normalized_labels:
        rows["h0"], rows["w0"] = img_size0
    rows["h1"], rows["w1"] = img_size
    rows["padh"], rows["padw"] = use.pad
    rows["bx2"] = rows["cx2"] = -1.0
    if use.border is not None:
        rows["bx1"], rows["bx2"], rows["by1"], rows["by2"] = use.border
    if use.crop is not None:
        rows["cx1"], rows["cx2"], rows["cy1"], rows["cy2"] = use.crop
    return rows


# ------------------------------------------------------------------------------------------------ pixels (device), single-stage wrappers
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
    dw, dh = (new_shape[1] - new_unpad[0]) / 2, (new_shape[0] - new_unpad[1]) / 2
    edges = tuple(int(round(v)) for v in (dh - 0.1, dh + 0.1, dw - 0.1, dw + 0.1))
    return new_unpad, edges, (dh, dw)


def pad_to_square(img, new_shape, pad_value=114):
    """Device letterbox: img [H, W, 3] uint8 -> (canvas [top + NH + bottom, left + NW + right, 3], (dh, dw)) as the reference returns."""
    hip.require_device(img, "pad_to_square")
    H, W, _ = img.shape
    (nw, nh), (top, bottom, left, right), pad = pad_to_square_plan((H, W), new_shape)
    OH, OW = top + nh + bottom, left + nw + right
    out = torch.empty((OH, OW, 3), dtype=torch.uint8, device=img.device)
    hip.call("ryolo_letterbox_u8", hip.ptr(img.contiguous()), H, W, nh, nw, top, left, hip.ptr(out), OH, OW, pad_value, hip.stream())
    return out, pad
