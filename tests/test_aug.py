"""SURVEY.md §8(f) N2 — mosaic-4 / mosaic-9 / mixup / warp / hsv of the reference's loader.
CPU: the package's placement tables (datasets/augment.py) replayed with the random draws recorded in fixture G11 (the real reference
load_mosaic / load_mosaic9 / load_target / mixup / random_warping ran: tests/golden/make_golden_aug.py) -> the rectangles reproduce the
reference's canvases when pasted with numpy, the warp matrix reproduces its labels.
GPU: the same canvases from the device paste kernel and the device mixup, BIT-EXACT against the fixture; the labels from the device
label stage (mosaic: bit-exact); warp, hsv, resize against this build's numpy restatement of OpenCV (parity unpinned: OpenCV is absent
and un-versioned in the reference)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_aug.npz"))
S = int(G["S"])
NIMG = int(G["nimg"])


def _images():
    return [G[f"img{i}"] for i in range(NIMG)]


def _np_paste(images, uses, CH, CW):
    cv = np.full((CH, CW, 3), 114, np.uint8)
    for u in uses:
        r = u.rect
        if r.w > 0 and r.h > 0:
            cv[r.dy:r.dy + r.h, r.dx:r.dx + r.w] = images[u.img][r.sy:r.sy + r.h, r.sx:r.sx + r.w]
    return cv


def _uses(kind, case):
    from ryolov4_amd.datasets import augment as A
    images = _images()
    idx = [int(i) for i in G[f"{kind}_{case}_idx"]]
    yc, xc = [int(v) for v in G[f"{kind}_{case}_yc_xc"]]
    shapes = [images[i].shape[:2] for i in idx]
    return (A.mosaic4_uses if kind == "m4" else A.mosaic9_uses)(shapes, idx, S, yc, xc), idx


@pytest.mark.parametrize("case", range(4))
def test_mosaic_placements_replay_the_reference(case):
    """The placement tables (MOSAIC4_CORNER / MOSAIC9_ORIGIN + window clipping) pasted with numpy reproduce the canvases the imported
    reference's load_mosaic / load_mosaic9 built (fixture G11)."""
    images = _images()
    for kind in ("m4", "m9"):
        uses, _ = _uses(kind, case)
        assert np.array_equal(_np_paste(images, uses, 2 * S, 2 * S), G[f"{kind}_{case}_img"]), kind


@pytest.mark.parametrize("case", range(3))
def test_warp_matrix_replays_the_reference(case):
    """warp_matrix for the four recorded draws: the fixture's warped labels (the reference's own double matrix product) follow from it."""
    from ryolov4_amd.datasets import augment as A
    a, s, tx, ty = [float(v) for v in G[f"warp_{case}_draws"]]
    M, (w, h) = A.warp_matrix((2 * S, 2 * S), a, s, tx, ty, border=(-S // 2, -S // 2))
    assert (w, h) == (S, S) and np.array_equal(M[2], [0.0, 0.0, 1.0])
    lab = G[f"m4_{case}_labels"].astype(np.float64)
    pts = np.concatenate((lab[:, 2:].reshape(-1, 2), np.ones((lab.shape[0] * 4, 1))), 1)
    got = (pts @ M.T)[:, :2].reshape(-1, 8).astype(np.float32)
    np.testing.assert_allclose(got, G[f"warp_{case}_labels"][:, 2:], rtol=1e-6, atol=1e-5)


def _label_rows(kind, case, mat=-1):
    from ryolov4_amd.datasets import augment as A
    uses, idx = _uses(kind, case)
    rows = []
    for u, i in zip(uses, idx):
        hw = G[f"img{i}"].shape[:2]
        rows.append(A.label_rows(G[f"polys{i}"], G[f"labels{i}"], 0, hw, hw, u, mat, normalized_labels=True))
    return np.concatenate(rows)


def _survivors(t):
    t = t.cpu().numpy()
    return t[~np.isnan(t[:, 2])]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(4))
def test_device_label_stage_replays_the_reference(case):
    """load_target / mosaic-9 crop / vertex warp of every label row in ONE element-wise launch: surviving rows, in order, equal the
    labels the imported reference produced (fixture G11) — mosaic labels to the last bit, warped ones within 1e-5."""
    from ryolov4_amd.datasets import augment as A
    dev = torch.device("cuda:0")
    for kind in ("m4", "m9"):
        got = _survivors(A.label_stage(_label_rows(kind, case), None, dev))
        want = G[f"{kind}_{case}_labels"]
        assert got.shape == want.shape and np.array_equal(got[:, 1:], want[:, 1:]), (kind, case)
    if case < 3:
        a, s, tx, ty = [float(v) for v in G[f"warp_{case}_draws"]]
        M, _ = A.warp_matrix((2 * S, 2 * S), a, s, tx, ty, border=(-S // 2, -S // 2))
        got = _survivors(A.label_stage(_label_rows("m4", case, mat=0), M[None], dev))
        np.testing.assert_allclose(got[:, 1:], G[f"warp_{case}_labels"][:, 1:], rtol=1e-6, atol=1e-5)


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
        rects += A.pool_rects(pool, _uses("m4", case)[0], canvas=2 * case)
        rects += A.pool_rects(pool, _uses("m9", case)[0], canvas=2 * case + 1)
        want += [G[f"m4_{case}_img"], G[f"m9_{case}_img"]]
    canv = A.paste(pool.buf, rects, 8, 2 * S, 2 * S)                                   # ONE launch for the eight canvases
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
@pytest.mark.parametrize("shape,new", [((48, 80), (64, 64)), ((64, 64), (64, 64)), ((30, 21), (96, 96)), ((100, 37), (64, 64)),
                                       ((128, 100), (64, 64))])          # last: exactly half size (OpenCV's INTER_LINEAR -> 2 x 2 block mean)
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


@pytest.mark.gpu
def test_device_resize_hsv_batch_against_numpy_restatement():
    """load_image's resize (+ hsv) for a whole batch of source images in one launch: INTER_LINEAR, INTER_AREA (eval loader, r < 1), and
    the r == 1 copy, with and without the hsv tables — equal to the numpy restatements bit for bit (parity unpinned: OpenCV restated)."""
    from oracle import ref_data
    from ryolov4_amd.datasets import augment as A
    rs = np.random.RandomState(5)
    images = [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in ((40, 64), (64, 48), (33, 61), (16, 16), (50, 37), (45, 60))]
    pool = A.ImagePool(images, torch.device("cuda:0"))
    gains = [(1.01, 1.4, 0.7), (0.99, 0.5, 1.3)]
    luts = np.stack([A.hsv_luts(np.asarray(g, dtype=np.float64)) for g in gains])
    items = [(0, (20, 32), A.INTERP_LINEAR, -1), (1, (32, 24), A.INTERP_AREA, -1), (2, (17, 32), A.INTERP_AREA, 0), (3, (16, 16), A.INTERP_COPY, 1),
             (4, (64, 47), A.INTERP_LINEAR, 1), (0, (13, 21), A.INTERP_AREA, -1), (1, (32, 24), A.INTERP_LINEAR, 0),
             # whole-number scale factors (cv::resizeAreaFast_): 3 x 3, 4 x 4 and 2 x 4 block sums; INTER_LINEAR at 3 x 3 stays bilinear
             (5, (15, 20), A.INTERP_AREA, -1), (1, (16, 12), A.INTERP_AREA, 1), (1, (32, 12), A.INTERP_AREA, -1), (5, (15, 20), A.INTERP_LINEAR, -1)]
    stage, offs = A.resize_hsv_batch(pool, items, luts)
    for (img, (nh, nw), interp, lut), off in zip(items, offs):
        src = images[img]
        ref = src if interp == A.INTERP_COPY else (ref_data.resize_area_numpy if interp == A.INTERP_AREA else ref_data.resize_linear_numpy)(src, (nw, nh))
        if lut >= 0:
            ref = ref_data.hsv_gain_numpy(ref, gains[lut])
        got = stage[off:off + nh * nw * 3].view(nh, nw, 3).cpu().numpy()
        assert np.array_equal(got, ref), (img, nh, nw, interp, lut, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_whole_number_area_resize_known_answers():
    """cv::resizeAreaFast_ restated (oracle/ref_data.py): 2 x 2 blocks round half UP ((sum + 2) >> 2), other whole-number blocks go through a
    float product rounded half to EVEN; cv::resize switches INTER_LINEAR to that path at exactly 2 x 2 and nowhere else."""
    from oracle import ref_data
    a = np.array([[0, 0, 1, 1], [0, 1, 0, 0], [255, 255, 3, 3], [255, 254, 3, 2]], dtype=np.uint8)[:, :, None].repeat(3, 2)
    exp = np.array([[0, 1], [255, 3]], dtype=np.uint8)                # sums 1, 2 (tie -> up), 1019, 11
    assert np.array_equal(ref_data.resize_area_numpy(a, (2, 2))[:, :, 0], exp)
    assert np.array_equal(ref_data.resize_linear_numpy(a, (2, 2))[:, :, 0], exp)
    b = np.zeros((4, 8, 3), dtype=np.uint8)
    b[:, :4] = 1                                                       # 4 x 4 block of ones -> 1; second block: a single 8 -> sum 8 / 16 = 0.5 -> even -> 0
    b[0, 4] = 8
    assert ref_data.area_fast_scales((4, 8), (2, 1)) == (4, 4)
    assert ref_data.resize_area_numpy(b, (2, 1))[0, :, 0].tolist() == [1, 0]
    b[0, 5] = 16                                                       # sum 24 / 16 = 1.5 -> even -> 2
    assert ref_data.resize_area_numpy(b, (2, 1))[0, :, 0].tolist() == [1, 2]
    assert ref_data.area_fast_scales((1601, 1200), (600, 800)) is None and ref_data.area_fast_scales((1600, 1200), (600, 800)) == (2, 2)
    assert ref_data.area_fast_scales((64, 64), (64, 64)) is None       # r == 1 never reaches cv2.resize; not a "downscale" here either
