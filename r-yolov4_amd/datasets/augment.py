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
    """Decoded uint8 HWC (BGR) images of different sizes (what cv2.imread leaves per file) resident in HBM, filled ON DEMAND and bounded
    by a byte budget — the role of the reference's 8 DataLoader workers re-reading files from disk (lib/load.py:19).

    Memory is a list of SLABS (one torch uint8 tensor each); an image is bump-allocated into the current slab and addressed by its 64-bit
    byte offset from a small anchor tensor (`buf`), which is what the table-driven kernels take (ryolo_resize_hsv_batch / ryolo_paste_rects:
    base + int64 offset, positive or negative).  Without a budget slabs are 1 GiB.  With `budget_bytes` a slab holds ONE image by default
    (`slab_bytes` = the image's size; a mosaic batch touches images all over the dataset, so coarser slabs would all be pinned by the batch
    being assembled), the sum of slab sizes never exceeds the budget, and when room is needed the LEAST RECENTLY USED slabs that the current
    batch does not need are dropped (their images leave the index; a dropped slab of sufficient size is reused in place, otherwise its memory
    goes back to torch's caching allocator).  A dropped image is decoded again when it is next asked for.  Reuse is ordered by the stream:
    the upload that overwrites a slab is enqueued on the same stream as the kernels that read it before.

    Host side: a decoded image is copied once into a PINNED staging tensor (torch's caching host allocator: the block is recycled only
    after the asynchronous copy that reads it has completed) and uploaded with a non-blocking copy; missing images of a batch are decoded
    by a thread pool (`workers`), so the host never holds more than the images in flight.

    ImagePool(images, device) — the round-2/3 form — is the eager special case: every image is uploaded at construction, no budget."""

    def __init__(self, images=None, device=None, *, count=None, decode=None, shapes=None, budget_bytes=None, slab_bytes=None, workers=8):
        self.device = torch.device(device)
        if images is not None:
            images = list(images)
            count, shapes = len(images), [tuple(im.shape[:2]) for im in images]
            decode = images.__getitem__
        self.count, self._decode, self.workers = int(count), decode, max(1, int(workers))
        self._shapes = {} if shapes is None else {i: tuple(s) for i, s in enumerate(shapes)}
        self.budget_bytes = None if budget_bytes is None else int(budget_bytes)
        if slab_bytes is None and images is not None:               # eager pools are small (tests, fixtures): one slab of their own size
            slab_bytes = max(sum(((h * w * 3 + 15) // 16) * 16 for h, w in shapes), 16)
        if slab_bytes is None and self.budget_bytes is None:
            slab_bytes = 1 << 30
        self.slab_bytes = None if slab_bytes is None else int(slab_bytes)     # None (budgeted pools): one image per slab
        self._anchor = None
        self._slabs, self._members, self._fill, self._touched = [], [], [], []     # per slab: tensor (None: freed), image ids, bump pointer, LRU tick
        self._where = {}                                                            # image id -> (slab, offset inside the slab)
        self._tick, self._cur, self._resident = 0, -1, 0
        self._executor = None
        self.stats = {"decoded": 0, "uploaded_bytes": 0, "evicted_slabs": 0, "hits": 0}
        if images is not None:
            self.ensure(range(self.count))

    # ---- what the kernels see
    @property
    def buf(self):
        if self._anchor is None:
            self._anchor = torch.zeros(16, dtype=torch.uint8, device=self.device)
        return self._anchor

    @property
    def shapes(self):
        return _ShapeView(self)

    @property
    def offsets(self):
        return _OffsetView(self)

    def shape(self, i):
        """(h, w) of image i without needing its pixels resident (known from the constructor or remembered from the first decode)."""
        if i not in self._shapes:
            self.ensure([i])
        return self._shapes[i]

    def offset(self, i):
        """Byte offset of RESIDENT image i relative to `buf` (call ensure() for the batch first)."""
        k, off = self._where[i]
        return self._slabs[k].data_ptr() - self.buf.data_ptr() + off

    def resident_bytes(self):
        return self._resident

    # ---- residency
    def _slab_for(self, nbytes, pinned):
        if self._cur >= 0 and self._slabs[self._cur] is not None and self._fill[self._cur] + nbytes <= self._slabs[self._cur].numel():
            return self._cur
        want = max(self.slab_bytes or 0, nbytes)
        if self.budget_bytes is not None:
            if want > self.budget_bytes:
                raise RuntimeError(f"ImagePool: one image needs {want} bytes, the budget is {self.budget_bytes}")
            while self._resident + want > self.budget_bytes:
                victims = [k for k, t in enumerate(self._slabs) if t is not None and k not in pinned]
                if not victims:
                    raise RuntimeError(f"ImagePool: the images of ONE batch do not fit the budget ({self.budget_bytes} bytes, {self._resident} "
                                       "held by this batch); raise pool_budget_bytes or lower the batch size")
                k = min(victims, key=lambda v: self._touched[v])
                for i in self._members[k]:
                    del self._where[i]
                self._members[k], self._fill[k] = [], 0
                self.stats["evicted_slabs"] += 1
                if self._slabs[k].numel() >= want:                  # reuse in place
                    self._cur = k
                    return k
                self._resident -= self._slabs[k].numel()
                self._slabs[k] = None                               # back to the caching allocator (stream-ordered reuse)
        self._slabs.append(torch.empty(want, dtype=torch.uint8, device=self.device))
        self._members.append([])
        self._fill.append(0)
        self._touched.append(self._tick)
        self._resident += want
        self._cur = len(self._slabs) - 1
        return self._cur

    def ensure(self, indices):
        """Make every image of `indices` resident (decode + upload the missing ones on the current stream); the slabs holding them are
        protected from eviction for the duration of this call, i.e. one call = one batch."""
        self._tick += 1
        need = list(dict.fromkeys(int(i) for i in indices))
        pinned = set()
        missing = []
        for i in need:
            w = self._where.get(i)
            if w is None:
                missing.append(i)
            else:
                pinned.add(w[0])
                self._touched[w[0]] = self._tick
                self.stats["hits"] += 1
        if not missing:
            return
        if len(missing) > 1 and self.workers > 1:
            if self._executor is None:
                from concurrent.futures import ThreadPoolExecutor
                self._executor = ThreadPoolExecutor(self.workers)
            decoded = self._executor.map(self._decode_one, missing)
        else:
            decoded = map(self._decode_one, missing)
        for i, stage in zip(missing, decoded):
            nbytes = stage.numel()
            k = self._slab_for(((nbytes + 15) // 16) * 16, pinned)
            pinned.add(k)
            off = self._fill[k]
            self._slabs[k][off:off + nbytes].copy_(stage, non_blocking=self.device.type == "cuda")
            self._fill[k] = off + ((nbytes + 15) // 16) * 16
            self._members[k].append(i)
            self._where[i] = (k, off)
            self._touched[k] = self._tick
            self.stats["decoded"] += 1
            self.stats["uploaded_bytes"] += nbytes

    def _decode_one(self, i):
        img = np.asarray(self._decode(i))
        if img.ndim == 2:
            img = img[:, :, None]
        if img.shape[2] != 3:                                       # base_dataset.py:177-178: grey images are stacked to 3 channels
            img = np.repeat(img[:, :, :1], 3, axis=2)
        self._shapes[i] = tuple(img.shape[:2])
        # straight into PINNED staging (inside the worker thread: numpy copies release the GIL), so the host holds one copy per image in flight
        stage = torch.empty(img.shape[0] * img.shape[1] * 3, dtype=torch.uint8, pin_memory=self.device.type == "cuda" and torch.cuda.is_available())
        np.copyto(stage.numpy().reshape(img.shape), img, casting="unsafe")
        return stage


class _ShapeView:
    def __init__(self, pool):
        self.pool = pool

    def __getitem__(self, i):
        return self.pool.shape(i)

    def __len__(self):
        return self.pool.count


class _OffsetView:
    def __init__(self, pool):
        self.pool = pool

    def __getitem__(self, i):
        return self.pool.offset(i)


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
    """A small host table (ctypes array / numpy array) -> device bytes through PINNED memory with a non-blocking copy on the current
    stream (the pinned block comes from torch's caching host allocator, which recycles it only after the copy has completed)."""
    raw = np.frombuffer(arr, dtype=np.uint8) if not isinstance(arr, np.ndarray) else np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    dev = torch.device(dev)
    if dev.type != "cuda":
        return torch.from_numpy(raw.copy()).to(dev)
    stage = torch.empty(max(raw.size, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
    stage.numpy()[:raw.size] = raw
    return stage[:max(raw.size, 1)].to(dev, non_blocking=True)


def paste(src_buf, rects, ncanvas, CH, CW, fill=114):
    """rects: [(source byte offset, source row pitch in pixels, Placed, canvas index)] in paste order (later rectangles win) ->
    canvases [ncanvas, CH, CW, 3] uint8, `fill` where nothing was pasted.  Rectangles listed canvas by canvas (what the batch assembler
    produces) take the grouped entry point: a pixel then looks at its own canvas's rectangles only."""
    _check_layouts()
    live = [(off, pitch, r, cv) for off, pitch, r, cv in rects if r.w > 0 and r.h > 0]         # empty numpy slices paste nothing
    arr = (_Rect * max(len(live), 1))()
    for k, (off, pitch, r, cv) in enumerate(live):
        arr[k] = _Rect(off, pitch, r.sx, r.sy, r.dx, r.dy, r.w, r.h, cv)
    dev = src_buf.device
    table = _to_device(arr, dev)
    canvas = torch.empty((ncanvas, CH, CW, 3), dtype=torch.uint8, device=dev)
    cvs = [cv for _, _, _, cv in live]
    if all(a <= b for a, b in zip(cvs, cvs[1:])) and (not cvs or (0 <= cvs[0] and cvs[-1] < ncanvas)):
        first = np.searchsorted(np.asarray(cvs, dtype=np.int64), np.arange(ncanvas + 1)).astype(np.int32)
        hip.call("ryolo_paste_rects_grouped", hip.ptr(src_buf), hip.ptr(table), len(live), hip.ptr(_to_device(first, dev)), hip.ptr(canvas), ncanvas, CH, CW,
                 fill, hip.stream())
    else:
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
        lt = None if luts is None or not len(luts) else _to_device(np.ascontiguousarray(luts, dtype=np.uint8), dev)
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
        table = _to_device(np.frombuffer(rows.tobytes(), dtype=np.uint8), device)
        mt = None if mats is None or not len(mats) else _to_device(np.ascontiguousarray(mats, dtype=np.float64).reshape(-1, 9), device).view(torch.float64)
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
    minv = _to_device(np.stack([np.linalg.inv(np.asarray(M, dtype=np.float64)) for M in Ms]).reshape(B, 9), imgs.device).view(torch.float64)
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
