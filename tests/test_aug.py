"""SURVEY.md §8(f) N2 — mosaic-4 / mosaic-9 / mixup / warp / hsv of the reference's loader.
CPU: the package's host-side plan and label code (datasets/augment.py) replayed with the random draws recorded in fixture G11 (the
real reference load_mosaic / load_mosaic9 / load_target / mixup / random_warping ran: tests/golden/make_golden_aug.py) -> labels equal
to the last bit, rectangles reproduce the reference's canvases when pasted with numpy.
GPU: the same canvases from the device paste kernel and the device mixup, BIT-EXACT against the fixture; warp and hsv against this
build's numpy restatement of OpenCV (parity unpinned: OpenCV is absent and un-versioned in the reference)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_aug.npz"))
S = int(G["S"])
NIMG = int(G["nimg"])


def _images():
    return [G[f"img{i}"] for i in range(NIMG)]


class _HostPool:                      # shapes / offsets only: enough for the plan code on CPU
    def __init__(self, images):
        self.shapes = [tuple(im.shape[:2]) for im in images]


def _np_paste(images, rects, CH, CW):
    cv = np.full((CH, CW, 3), 114, np.uint8)
    for (img, sx, sy, dx, dy, w, h, _) in rects:
        if w > 0 and h > 0:
            cv[dy:dy + h, dx:dx + w] = images[img][sy:sy + h, sx:sx + w]
    return cv


def _labels(kind, case, meta, idx, xc=0, yc=0):
    from ryolov4_amd.datasets import augment as A
    out = []
    for i, (pad, boarder) in zip(idx, meta):
        im = G[f"img{i}"]
        out.append(A.load_target(torch.from_numpy(G[f"polys{i}"].copy()), torch.from_numpy(G[f"labels{i}"].copy()), pad,
                                 im.shape[:2], im.shape[:2], True, boarder=boarder))
    lab = torch.cat(out, 0)
    if kind == "m9":
        lab = A.filtering(lab, (xc, xc + 2 * S, yc, yc + 2 * S))
        lab[:, [2, 4, 6, 8]] -= xc
        lab[:, [3, 5, 7, 9]] -= yc
    return lab


@pytest.mark.parametrize("case", range(4))
def test_mosaic_plans_and_labels_replay_the_reference(case):
    from ryolov4_amd.datasets import augment as A
    images = _images()
    pool = _HostPool(images)
    idx = [int(i) for i in G[f"m4_{case}_idx"]]
    yc, xc = [int(v) for v in G[f"m4_{case}_yc_xc"]]
    rects, meta = A.mosaic4(pool, idx, S, yc, xc)
    assert np.array_equal(_np_paste(images, rects, 2 * S, 2 * S), G[f"m4_{case}_img"])
    assert torch.equal(_labels("m4", case, meta, idx), torch.from_numpy(G[f"m4_{case}_labels"]))
    idx9 = [int(i) for i in G[f"m9_{case}_idx"]]
    yc9, xc9 = [int(v) for v in G[f"m9_{case}_yc_xc"]]
    rects9, meta9 = A.mosaic9(pool, idx9, S, yc9, xc9)
    assert np.array_equal(_np_paste(images, rects9, 2 * S, 2 * S), G[f"m9_{case}_img"])
    assert torch.equal(_labels("m9", case, meta9, idx9, xc9, yc9), torch.from_numpy(G[f"m9_{case}_labels"]))


@pytest.mark.parametrize("case", range(3))
def test_warp_labels_replay_the_reference(case):
    from ryolov4_amd.datasets import augment as A
    a, s, tx, ty = [float(v) for v in G[f"warp_{case}_draws"]]
    M, (w, h) = A.warp_matrix((2 * S, 2 * S), a, s, tx, ty, border=(-S // 2, -S // 2))
    assert (w, h) == (S, S)
    tg = A.warp_targets(torch.from_numpy(G[f"m4_{case}_labels"].copy()), M)
    assert torch.equal(tg, torch.from_numpy(G[f"warp_{case}_labels"]))


def test_oracle_restatements_against_fixture():
    from oracle import ref_data
    for case in range(2):
        a, b = G[f"m4_{case}_img"], G[f"m4_{case + 2}_img"]
        assert np.array_equal(ref_data.mixup_numpy(a, b, float(G[f"mix_{case}_r"])), G[f"mix_{case}_img"])
    ident = ref_data.warp_perspective_numpy(G["m4_0_img"], np.eye(3), (2 * S, 2 * S))
    assert np.array_equal(ident, G["m4_0_img"])                                      # identity map: every weight table entry is (32768, 0, 0, 0)
    shifted = ref_data.warp_perspective_numpy(G["m4_0_img"], np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1]]), (2 * S, 2 * S))
    assert np.array_equal(shifted[:-2, 3:], G["m4_0_img"][2:, :-3]) and (shifted[:, :3] == 114).all()


@pytest.mark.gpu
def test_device_paste_and_mixup_bit_exact():
    from ryolov4_amd.datasets import augment as A
    images = _images()
    pool = A.ImagePool(images, torch.device("cuda:0"))
    rects, want = [], []
    for case in range(4):
        idx = [int(i) for i in G[f"m4_{case}_idx"]]
        yc, xc = [int(v) for v in G[f"m4_{case}_yc_xc"]]
        rects += A.mosaic4(pool, idx, S, yc, xc, canvas=2 * case)[0]
        idx9 = [int(i) for i in G[f"m9_{case}_idx"]]
        yc9, xc9 = [int(v) for v in G[f"m9_{case}_yc_xc"]]
        rects += A.mosaic9(pool, idx9, S, yc9, xc9, canvas=2 * case + 1)[0]
        want += [G[f"m4_{case}_img"], G[f"m9_{case}_img"]]
    canv = A.paste(pool, rects, 8, 2 * S, 2 * S)                                       # ONE launch for the eight canvases
    for k, w in enumerate(want):
        assert np.array_equal(canv[k].cpu().numpy(), w), k
    for case in range(2):
        mixed = A.mixup(canv[2 * case], canv[2 * (case + 2)], float(G[f"mix_{case}_r"]))
        assert np.array_equal(mixed.cpu().numpy(), G[f"mix_{case}_img"])


@pytest.mark.gpu
def test_device_warp_and_hsv_against_numpy_restatement():
    """Parity unpinned (OpenCV restated): the device kernels equal the numpy restatement bit for bit."""
    from oracle import ref_data
    from ryolov4_amd.datasets import augment as A
    imgs = np.stack([G[f"m4_{c}_img"] for c in range(3)])
    Ms = []
    for case in range(3):
        a, s, tx, ty = [float(v) for v in G[f"warp_{case}_draws"]]
        Ms.append(A.warp_matrix((2 * S, 2 * S), a, s, tx, ty, border=(-S // 2, -S // 2))[0])
    dev = torch.from_numpy(imgs).cuda()
    out = A.warp_perspective(dev, Ms, (S, S)).cpu().numpy()
    for case in range(3):
        ref = ref_data.warp_perspective_numpy(imgs[case], Ms[case], (S, S))
        assert np.array_equal(out[case], ref), (case, int(np.abs(out[case].astype(int) - ref.astype(int)).max()))
        assert np.array_equal(ref, G[f"warp_{case}_img_unpinned"])
    for r in ((1.0, 1.0, 1.0), (1.01, 1.4, 0.7), (0.99, 0.5, 1.3)):
        got = A.hsv_gain(dev.clone(), r).cpu().numpy()
        ref = ref_data.hsv_gain_numpy(imgs, r)
        assert np.array_equal(got, ref), (r, int(np.abs(got.astype(int) - ref.astype(int)).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,new", [((48, 80), (64, 64)), ((64, 64), (64, 64)), ((30, 21), (96, 96)), ((100, 37), (64, 64))])
def test_device_letterbox_against_numpy_restatement(shape, new):
    """pad_to_square (datasets/base_dataset.py:33-56): integer plan shared with the reference's arithmetic; resize pixels restate
    OpenCV's 8-bit INTER_LINEAR (parity unpinned) and equal the numpy restatement bit for bit; the border is 114."""
    from oracle import ref_data
    from ryolov4_amd.datasets import augment as A
    img = np.random.RandomState(shape[0]).randint(0, 256, size=shape + (3,)).astype(np.uint8)
    ref, pad_ref = ref_data.pad_to_square_numpy(img, new)
    got, pad = A.pad_to_square(torch.from_numpy(img).cuda(), new)
    assert pad == pad_ref and tuple(got.shape) == ref.shape
    assert np.array_equal(got.cpu().numpy(), ref)
