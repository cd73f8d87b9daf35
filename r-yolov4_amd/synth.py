"""Synthetic inputs of SURVEY.md §8(d): training batches, rotated-box targets, NMS candidate sets, and the
closed-form weight fill used by the golden fixtures.  Pure numpy/torch-CPU; no dependency on the HIP library."""
import math

import numpy as np
import torch

# data/hyp.yaml:3-7 and :12-17 of the reference (values are part of the parity contract)
CFG = {"anchors": [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146], [142, 110, 192, 243, 459, 401]],
       "angles": [-90, -60, -30, 0, 30, 60]}
HYP = {"fl_gamma": 0.0, "box": 0.05, "obj": 1.0, "obj_pw": 1.0, "cls": 0.5, "cls_pw": 1.0}


def fill_state(sd):
    """Deterministic closed-form fill of a state_dict: value = f(key order, flat index).  Regenerated identically
    on the GPU box, so no reference weights ever travel."""
    out = {}
    for k, (name, t) in enumerate(sd.items()):
        n = t.numel()
        i = torch.arange(n, dtype=torch.float64)
        if name.endswith("num_batches_tracked"):
            v = torch.zeros(n, dtype=torch.float64)
        elif name.endswith("running_var"):
            v = 0.8 + 0.4 * torch.sin(0.31 * i + 0.7 * k) ** 2
        elif name.endswith("running_mean"):
            v = 0.05 * torch.sin(0.17 * i + 0.3 * k)
        elif t.dim() == 1 and name.endswith("weight"):          # BN gamma
            v = 1.0 + 0.1 * torch.sin(0.23 * i + 0.5 * k)
        elif name.endswith("bias"):
            v = 0.05 * torch.cos(0.19 * i + 0.9 * k)
        elif name.endswith("implicit"):
            base = 1.0 if ".im" in name else 0.0
            v = base + 0.02 * torch.sin(0.29 * i + 0.4 * k)
        else:                                                    # conv weight: hash-uniform, variance 0.81/fan_in (well conditioned)
            fan_in = max(1, n // t.shape[0])
            h = torch.sin(i * 12.9898 + k * 78.233 + 0.5) * 43758.5453
            v = (2.0 * (h - torch.floor(h)) - 1.0) * (math.sqrt(3.0) * 0.9 / math.sqrt(fan_in))
        out[name] = v.to(t.dtype).reshape(t.shape)
    return out


def gaussian_label(angle, num_class=180, u=0, sig=6.0):
    x = np.arange(-num_class / 2, num_class / 2)
    y = np.exp(-(x - u) ** 2 / (2 * sig ** 2))
    k = int(num_class / 2 - angle)
    return np.concatenate([y[k:], y[:k]], axis=0)


def synth_targets(B, per_image, nc, csl, seed=42, img_size=800, edge_cases=False):
    """targets [nt, 7|187] = (img, cls, x, y, w, h, theta[, csl x180]); xywh normalised, h >= w (dataset invariant),
    theta in [-pi/2, pi/2), image index grouped ascending (datasets/base_dataset.py:161-167)."""
    rng = np.random.RandomState(seed)
    rows = []
    for b in range(B):
        for _ in range(per_image):
            h_px = math.exp(rng.uniform(math.log(16), math.log(256)))
            w_px = h_px / rng.uniform(1, 5)
            th = rng.uniform(-math.pi / 2, math.pi / 2)
            th = min(th, math.pi / 2 - 1e-4)
            rows.append([b, rng.randint(0, nc), rng.uniform(0.05, 0.95), rng.uniform(0.05, 0.95),
                         w_px / img_size, h_px / img_size, th])
    if edge_cases and rows:
        # cell-boundary / image-border positions: gxy % 1 == 0.5 exactly, coords near 0 and near 1
        rows[0][2:4] = [0.5 + 0.5 / 8, 0.25 + 0.5 / 8]
        rows[1][2:4] = [0.003, 0.997]
        if len(rows) > 2:
            rows[2][2:4] = [0.999, 0.001]
    t = torch.tensor(rows, dtype=torch.float32).reshape(-1, 7)
    if csl:
        lab = [gaussian_label(float(r[6]) * 180 / np.pi + 90) for r in t]
        lab = torch.from_numpy(np.stack(lab)).float() if len(lab) else torch.zeros(0, 180)
        t = torch.cat((t, lab), 1)
    return t


def synth_batch(B, S, nc, csl, seed=42, per_image=64):
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(B, 3, S, S, generator=g)
    return imgs, synth_targets(B, per_image, nc, csl, seed=seed, img_size=S)


def synth_nms_boxes(n, dist="U", seed=0, ncls=16, max_wh=4096.0):
    """SURVEY §8(d) NMS inputs: boxes (xc,yc,w,h,angle_deg) with class offset, distinct scores, sorted desc."""
    rng = np.random.RandomState(seed)
    if dist == "U":
        xy = rng.uniform(0, 800, (n, 2))
        w = rng.uniform(8, 64, n)
        h = w * rng.uniform(1, 4, n)
        a = rng.uniform(-90, 90, n)
        c = rng.randint(0, ncls, n)
    else:   # clustered
        ns = max(1, n // 20)
        sxy = rng.uniform(0, 800, (ns, 2)); sw = rng.uniform(8, 64, ns); sh = sw * rng.uniform(1, 4, ns)
        sa = rng.uniform(-90, 90, ns); sc = rng.randint(0, ncls, ns)
        rep = np.arange(n) % ns
        xy = sxy[rep] + rng.normal(0, 2, (n, 2))
        w = sw[rep] * (1 + rng.normal(0, 0.05, n)); h = sh[rep] * (1 + rng.normal(0, 0.05, n))
        a = sa[rep] + rng.normal(0, 3, n); c = sc[rep]
    boxes = np.stack([xy[:, 0] + c * max_wh, xy[:, 1] + c * max_wh, np.abs(w), np.abs(h), a], 1).astype(np.float32)
    scores = rng.permutation(np.linspace(0.001, 0.999, n)).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    return boxes[order], scores[order]
