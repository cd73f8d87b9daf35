"""CPU restatement (torch-CPU / numpy) of the batch-finalisation end of the reference's data pipeline and of the detect path's
box geometry — SURVEY.md §8(f) N2 slice + N4.  TEST INFRASTRUCTURE ONLY: imported by tests/ and the golden generator, never by
the product path (r-yolov4_amd/).

Pinned against the imported reference by tests/golden/make_golden_data.py (fixture g9_data.npz): BaseDataset.__getitem__'s tail
and collate_fn run for real there (datasets/base_dataset.py:129-166), as do xyxyxyxy2xywha / xywha2xyxyxyxy (lib/general.py) and
rescale_boxes (lib/plot.py).  cv2 is absent in this image: `get_rotation_matrix_2d` restates OpenCV's documented closed form
(cv::getRotationMatrix2D) and is what the generator plugs into the reference's `cv.getRotationMatrix2D` call — that one call is
"parity unpinned" (third-party, un-pinned version); everything around it is the reference's own arithmetic.
"""
import math

import numpy as np
import torch

from .ref_ops import norm_angle


def gaussian_label(label, num_class=180, u=0, sig=6.0):
    """datasets/base_dataset.py:13-31.  `int()` truncates toward zero; negative indices wrap through python slicing."""
    x = np.arange(-num_class / 2, num_class / 2)
    y_sig = np.exp(-(x - u) ** 2 / (2 * sig ** 2))
    index = int(num_class / 2 - label)
    return np.concatenate([y_sig[index:], y_sig[:index]], axis=0)


def filtering(targets, border):
    """datasets/base_dataset.py:340-352: keep polygons whose vertex mean lies strictly inside (x1, x2, y1, y2)."""
    x1, x2, y1, y2 = border
    x = torch.mean(targets[:, [2, 4, 6, 8]], dim=1)
    y = torch.mean(targets[:, [3, 5, 7, 9]], dim=1)
    return targets[(x > x1) & (x < x2) & (y > y1) & (y < y2)]


def normalize(targets, img_size):
    """datasets/base_dataset.py:354-361 (in place)."""
    height, width = img_size
    targets[:, [2, 4, 6, 8]] /= width
    targets[:, [3, 5, 7, 9]] /= height
    return targets


def horizontal_flip(image, targets):
    """lib/augmentations.py:39-42."""
    targets[:, [2, 4, 6, 8]] = 1 - targets[:, [2, 4, 6, 8]]
    return np.fliplr(image), targets


def vertical_flip(image, targets):
    """lib/augmentations.py:33-36."""
    targets[:, [3, 5, 7, 9]] = 1 - targets[:, [3, 5, 7, 9]]
    return np.flipud(image), targets


def xyxyxyxy2xywha(boxes):
    """lib/general.py:70-104: clockwise polygon -> (x, y, w, h, theta); h = long side, theta in [-pi/2, pi/2)."""
    x1, y1, x2, y2, x3, y3, x4, y4 = boxes.unbind(dim=-1)
    x = (x1 + x2 + x3 + x4) / 4
    y = (y1 + y2 + y3 + y4) / 4
    w = (torch.linalg.norm(torch.stack((x2 - x3, y2 - y3), -1), dim=1) + torch.linalg.norm(torch.stack((x1 - x4, y1 - y4), -1), dim=1)) / 2
    h = (torch.linalg.norm(torch.stack((x1 - x2, y1 - y2), -1), dim=1) + torch.linalg.norm(torch.stack((x4 - x3, y4 - y3), -1), dim=1)) / 2
    theta = -(torch.atan2(y1 - y2, x1 - x2) + torch.atan2(y4 - y3, x4 - x3)) / 2
    swap = w >= h                                                             # the reference loops per box (:92-99); same result
    w, h = torch.where(swap, h, w), torch.where(swap, w, h)
    theta = torch.where(swap, torch.where(theta > 0, theta - np.pi / 2, theta + np.pi / 2), theta)
    return torch.stack((x, y, w, h, norm_angle(theta)), -1)


def finalize_sample(img_bgr_u8, targets10, fliplr, flipud, csl):
    """datasets/base_dataset.py:129-157: filtering -> normalize -> flips -> poly->xywha (+ CSL) -> BGR->RGB CHW float / 255.
    targets10 [n, 10] = (0, cls, x1..y4) in pixels of the (already padded / warped) image."""
    img = img_bgr_u8
    targets = filtering(targets10.clone(), (0, img.shape[1], 0, img.shape[0]))
    targets = normalize(targets, img.shape[:2])
    if fliplr:
        img, targets = horizontal_flip(img, targets)
    if flipud:
        img, targets = vertical_flip(img, targets)
    labels = torch.zeros((0, 187 if csl else 7), dtype=torch.float32)
    if len(targets):
        rboxes = xyxyxyxy2xywha(targets[:, 2:])
        if csl:
            rows = [gaussian_label(label=rboxes[i, 4] * 180 / np.pi + 90, num_class=180, u=0, sig=6) for i in range(len(rboxes))]
            labels = torch.cat((targets[:, :2], rboxes, torch.from_numpy(np.stack(rows)).type(torch.float32)), -1)
        else:
            labels = torch.cat((targets[:, :2], rboxes), -1)
    t = np.ascontiguousarray(img.transpose((2, 0, 1))[::-1])
    return torch.from_numpy(t).float() / 255, labels


def collate(samples):
    """datasets/base_dataset.py:159-166 on a list of (img, labels)."""
    imgs, targets = list(zip(*samples))
    for i, boxes in enumerate(targets):
        boxes[:, 0] = i
    return torch.stack(imgs, 0), torch.cat(targets, 0)


# ------------------------------------------------------------------------------------------------ detect path
def get_rotation_matrix_2d(center, angle, scale):
    """OpenCV cv::getRotationMatrix2D (imgproc; documented closed form), float64 2x3:
    [[a, b, (1-a)cx - b cy], [-b, a, b cx + (1-a) cy]], a = scale cos(angle), b = scale sin(angle), angle in degrees."""
    ang = angle * math.pi / 180.0
    a, b = math.cos(ang) * scale, math.sin(ang) * scale
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def xywh2xyxy(x):
    """lib/general.py:23-38."""
    y = x.new(x.shape)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def xywha2xyxyxyxy(boxes):
    """lib/general.py:41-67 -> [N, 4, 2]; note h spans x and w spans y before the rotation (:59-62)."""
    n = boxes.size(0)
    Rs = torch.zeros((n, 2, 3))
    x, y, w, h, theta = boxes.unbind(dim=-1)
    for i in range(n):
        Rs[i] = torch.from_numpy(get_rotation_matrix_2d((float(x[i]), float(y[i])), float(theta[i] * 180 / np.pi), 1))
    p = torch.stack((x - h / 2, y - w / 2, x + h / 2, y - w / 2, x + h / 2, y + w / 2, x - h / 2, y + w / 2), dim=-1).reshape(-1, 4, 2)
    p = torch.cat((p, torch.ones((n, 4, 1))), dim=-1)
    return torch.bmm(p, Rs.permute((0, 2, 1)))


def rescale_boxes(boxes, current_dim, original_shape):
    """lib/plot.py:9-31 (in place on columns 0-3): undo pad-to-square + resize, xywh kept as centre/size."""
    orig_h, orig_w = original_shape
    pad_x = max(orig_h - orig_w, 0) * (current_dim / max(original_shape))
    pad_y = max(orig_w - orig_h, 0) * (current_dim / max(original_shape))
    unpad_h = current_dim - pad_y
    unpad_w = current_dim - pad_x
    boxes[:, :4] = xywh2xyxy(boxes[:, :4])
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    x1 = ((x1 - pad_x // 2) / unpad_w) * orig_w
    y1 = ((y1 - pad_y // 2) / unpad_h) * orig_h
    x2 = ((x2 - pad_x // 2) / unpad_w) * orig_w
    y2 = ((y2 - pad_y // 2) / unpad_h) * orig_h
    boxes[:, 0] = (x1 + x2) / 2
    boxes[:, 1] = (y1 + y2) / 2
    boxes[:, 2] = (x2 - x1)
    boxes[:, 3] = (y2 - y1)
    return boxes


# ---------------------------------------------------------------------------------------------- augmentations
def mosaic4_numpy(images, s, yc, xc):
    """datasets/base_dataset.py:224-268, image part: 4 uint8 HWC images -> 2s x 2s canvas (114-filled) + per image (pad, boarder).
    Pinned: tests/golden/make_golden_aug.py asserts equality with the reference's load_mosaic."""
    img4 = np.full((s * 2, s * 2, 3), 114, dtype=np.uint8)
    meta = []
    for i, img in enumerate(images):
        h, w = img.shape[:2]
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
        img4[y1a:y2a, x1a:x2a] = img[y1b:y2b, x1b:x2b]
        meta.append(((y1a - y1b, x1a - x1b), (x1b, x2b, y1b, y2b)))
    return img4, meta


def mixup_numpy(img, img2, r):
    """lib/augmentations.py:24-28 (image part)."""
    return (img * r + img2 * (1 - r)).astype(np.uint8)


def warp_perspective_numpy(src, M, dsize, border=114):
    """cv2.warpPerspective(src, M, dsize, borderValue = (114,)*3) restated from OpenCV's implementation (INTER_LINEAR: 5 fractional
    coordinate bits, 15-bit weight table whose rounding error goes to the largest / smallest weight, round-to-nearest result).
    PARITY UNPINNED: OpenCV is absent here and the reference does not pin its version."""
    DW, DH = dsize
    SH, SW = src.shape[:2]
    m = np.linalg.inv(np.asarray(M, dtype=np.float64)).reshape(9)
    xs, ys = np.meshgrid(np.arange(DW, dtype=np.float64), np.arange(DH, dtype=np.float64))
    W = m[6] * xs + m[7] * ys + m[8]
    W = np.where(W != 0, 32.0 / np.where(W != 0, W, 1.0), 0.0)
    X = np.rint(np.clip((m[0] * xs + m[1] * ys + m[2]) * W, -2**31, 2**31 - 1)).astype(np.int64)
    Y = np.rint(np.clip((m[3] * xs + m[4] * ys + m[5]) * W, -2**31, 2**31 - 1)).astype(np.int64)
    sx, sy, ax, ay = X >> 5, Y >> 5, X & 31, Y & 31
    fx1, fy1 = (ax.astype(np.float32) * np.float32(1 / 32)), (ay.astype(np.float32) * np.float32(1 / 32))
    wf = np.stack(((1 - fy1) * (1 - fx1), (1 - fy1) * fx1, fy1 * (1 - fx1), fy1 * fx1), -1).astype(np.float32)
    w = np.rint(wf * np.float32(32768)).astype(np.int64)
    diff = w.sum(-1) - 32768
    imax, imin = w.argmax(-1), w.argmin(-1)          # first maximum / first minimum, as the scan in initInterTab2D finds them
    ii, jj = np.indices(diff.shape)
    neg, pos = diff < 0, diff > 0
    w[ii[neg], jj[neg], imax[neg]] -= diff[neg]
    w[ii[pos], jj[pos], imin[pos]] -= diff[pos]
    out = np.zeros((DH, DW, 3), dtype=np.int64)
    for k in range(4):
        px, py = sx + (k & 1), sy + (k >> 1)
        ok = (px >= 0) & (px < SW) & (py >= 0) & (py < SH)
        v = np.where(ok[..., None], src[np.clip(py, 0, SH - 1), np.clip(px, 0, SW - 1)].astype(np.int64), border)
        out += v * w[..., k:k + 1]
    return ((out + (1 << 14)) >> 15).astype(np.uint8)


def bgr2hsv_numpy(img):
    """cv2.cvtColor(img, COLOR_BGR2HSV) for uint8, restated from OpenCV (integer path, 12-bit division tables, hue range 180).
    PARITY UNPINNED (OpenCV absent, version un-pinned)."""
    b, g, rr = [img[..., k].astype(np.int64) for k in range(3)]
    v = np.maximum(b, np.maximum(g, rr))
    vmin = np.minimum(b, np.minimum(g, rr))
    diff = v - vmin
    vr = np.where(v == rr, -1, 0)
    vg = np.where(v == g, -1, 0)
    sdiv = np.where(v > 0, np.rint((255 << 12) / np.maximum(v, 1).astype(np.float64)), 0).astype(np.int64)
    hdiv = np.where(diff > 0, np.rint((180 << 12) / (6.0 * np.maximum(diff, 1))), 0).astype(np.int64)
    s = (diff * sdiv + (1 << 11)) >> 12
    h = (vr & (g - b)) + (~vr & ((vg & (b - rr + 2 * diff)) + ((~vg) & (rr - g + 4 * diff))))
    h = (h * hdiv + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack((h & 255, s, v), -1).astype(np.uint8)


def hsv2bgr_numpy(hsv):
    """cv2.cvtColor(hsv, COLOR_HSV2BGR) for uint8: OpenCV's float converter on h * 2 degrees / s, v scaled by 1 / 255.  PARITY UNPINNED."""
    H, S, V = [hsv[..., k].astype(np.float32) for k in range(3)]
    hf, sf, vf = H * np.float32(6 / 180), S * np.float32(1 / 255), V * np.float32(1 / 255)
    hf = np.where(hf >= 6, hf - 6, hf)
    sec = np.floor(hf).astype(np.int64)
    fr = (hf - sec.astype(np.float32)).astype(np.float32)
    bad = (sec < 0) | (sec >= 6)
    sec = np.where(bad, 0, sec)
    fr = np.where(bad, np.float32(0), fr)
    tab = np.stack((vf, vf * (1 - sf), vf * (1 - sf * fr), vf * (1 - sf * (1 - fr))), -1).astype(np.float32)
    sector = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    out = np.take_along_axis(tab, sector[sec], -1)
    out = np.where((sf == 0)[..., None], vf[..., None], out)
    return np.clip(np.rint(out * np.float32(255)), 0, 255).astype(np.uint8)


def hsv_gain_numpy(img, r):
    """lib/augmentations.py:8-21 with cv2.cvtColor / cv2.LUT restated (bgr2hsv_numpy, hsv2bgr_numpy).  PARITY UNPINNED."""
    x = np.arange(0, 256, dtype=np.float64)
    lut_h = ((x * r[0]) % 180).astype(np.uint8)
    lut_s = np.clip(x * r[1], 0, 255).astype(np.uint8)
    lut_v = np.clip(x * r[2], 0, 255).astype(np.uint8)
    hsv = bgr2hsv_numpy(img)
    return hsv2bgr_numpy(np.stack((lut_h[hsv[..., 0]], lut_s[hsv[..., 1]], lut_v[hsv[..., 2]]), -1))


def area_fast_scales(src_hw, dsize):
    """cv::resize's `is_area_fast` test (imgproc/src/resize.cpp): scale = 1 / ((double) dst / src) per axis, iscale = saturate_cast<int>(scale)
    (round to nearest even), fast iff both |scale - iscale| < DBL_EPSILON.  Returns (iscale_x, iscale_y) for a whole-number DOWNSCALE, else None."""
    NW, NH = dsize
    SH, SW = src_hw
    if NW <= 0 or NH <= 0:
        return None
    sx, sy = 1.0 / (float(NW) / float(SW)), 1.0 / (float(NH) / float(SH))
    ix, iy = int(np.rint(sx)), int(np.rint(sy))
    eps = np.finfo(np.float64).eps
    if abs(sx - ix) < eps and abs(sy - iy) < eps and ix >= 1 and iy >= 1 and (ix > 1 or iy > 1):
        return ix, iy
    return None


def resize_area_fast_numpy(src, dsize, ix, iy):
    """OpenCV's whole-number INTER_AREA path for uint8 (cv::resizeAreaFast_): the iy x ix source block of a destination pixel is summed in
    int; 2 x 2 blocks take the vector form (sum + 2) >> 2 (round half UP), every other block saturate_cast<uchar>(sum * (1.f / area)) — a
    float product rounded half to EVEN.  (The generic float accumulation of resize_area_numpy differs from both in ties.)
    PARITY UNPINNED (OpenCV absent) — restated from cv::ResizeAreaFastVec / resizeAreaFast_Invoker."""
    NW, NH = dsize
    s = src[:NH * iy, :NW * ix].astype(np.int64).reshape(NH, iy, NW, ix, -1).sum(axis=(1, 3))
    if ix == 2 and iy == 2:
        return ((s + 2) >> 2).astype(np.uint8)
    scale = np.float32(1.0) / np.float32(ix * iy)
    return np.clip(np.rint(s.astype(np.float32) * scale), 0, 255).astype(np.uint8)


def resize_area_numpy(src, dsize):
    """cv2.resize(src, (w, h), interpolation = INTER_AREA) for uint8 DOWNSCALING, restated from OpenCV's generic cv::ResizeArea_: per
    axis every destination cell covers [d * scale, (d + 1) * scale) of the source — a leading partial source cell, whole cells of
    weight 1 / cellWidth, a trailing partial cell (weights in float); rows are accumulated horizontally, then vertically, in float;
    cvRound at the end.  Whole-number scale factors take OpenCV's integer path (resize_area_fast_numpy).  PARITY UNPINNED (OpenCV absent, version un-pinned by the reference)."""
    NW, NH = dsize
    SH, SW = src.shape[:2]
    fast = area_fast_scales((SH, SW), dsize)
    if fast is not None:
        return resize_area_fast_numpy(src, dsize, *fast)

    def axis(dn, sn):
        scale = sn / dn
        tabs = []
        for d in range(dn):
            f1 = d * scale
            f2 = f1 + scale
            cell = min(scale, sn - f1)
            s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
            s2 = min(s2, sn - 1)
            s1 = min(s1, s2)
            t = []
            if s1 - f1 > 1e-3:
                t.append((s1 - 1, np.float32((s1 - f1) / cell)))
            t += [(sx, np.float32(1.0 / cell)) for sx in range(s1, s2)]
            if f2 - s2 > 1e-3:
                t.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
            tabs.append(t)
        return tabs
    tx, ty = axis(NW, SW), axis(NH, SH)
    s = src.astype(np.float32)
    rows = np.zeros((SH, NW, 3), dtype=np.float32)
    for dx, t in enumerate(tx):
        acc = np.zeros((SH, 3), dtype=np.float32)
        for sx, w in t:
            acc = acc + s[:, sx] * w
        rows[:, dx] = acc
    out = np.zeros((NH, NW, 3), dtype=np.float32)
    for dy, t in enumerate(ty):
        acc = np.zeros((NW, 3), dtype=np.float32)
        for sy, w in t:
            acc = acc + w * rows[sy]
        out[dy] = acc
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_linear_numpy(src, dsize):
    """cv2.resize(src, (w, h), interpolation = INTER_LINEAR) for uint8, restated from OpenCV (11-bit coefficients, two-pass integer
    arithmetic).  PARITY UNPINNED (OpenCV absent, version un-pinned by the reference)."""
    NW, NH = dsize
    SH, SW = src.shape[:2]
    if area_fast_scales((SH, SW), dsize) == (2, 2):                # cv::resize: INTER_LINEAR at exactly 2x2 IS the INTER_AREA fast path
        return resize_area_fast_numpy(src, dsize, 2, 2)

    def coef(dn, sn):
        o = np.arange(dn, dtype=np.float64)
        f = ((o + 0.5) * (sn / dn) - 0.5).astype(np.float32)
        si = np.floor(f).astype(np.int64)
        f = (f - si.astype(np.float32)).astype(np.float32)
        lo, hi = si < 0, si >= sn - 1
        f = np.where(lo | hi, np.float32(0), f)
        si = np.where(lo, 0, np.where(hi, sn - 1, si))
        return si, np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64), np.rint(f * np.float32(2048)).astype(np.int64)
    sx, ax0, ax1 = coef(NW, SW)
    sy, by0, by1 = coef(NH, SH)
    sx1, sy1 = np.minimum(sx + 1, SW - 1), np.minimum(sy + 1, SH - 1)
    s = src.astype(np.int64)
    r0 = s[sy][:, sx] * ax0[None, :, None] + s[sy][:, sx1] * ax1[None, :, None]
    r1 = s[sy1][:, sx] * ax0[None, :, None] + s[sy1][:, sx1] * ax1[None, :, None]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def pad_to_square_numpy(img, new_shape, pad_value=114):
    """datasets/base_dataset.py:33-56 with cv2.resize -> resize_linear_numpy and cv2.copyMakeBorder(BORDER_CONSTANT) -> np.pad."""
    shape = img.shape[:2]
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        img = resize_linear_numpy(img, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return np.pad(img, ((top, bottom), (left, right), (0, 0)), constant_values=pad_value), (dh, dw)
