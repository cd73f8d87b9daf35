"""Device-side end of the reference's datasets/base_dataset.py (SURVEY §8(f) N2 slice): everything BaseDataset.__getitem__ does
AFTER the cv2 stages (mosaic / warp / hsv) plus collate_fn, for a whole batch in three launches.

    finalize_batch(imgs_u8, targets10, flags, csl)     base_dataset.py:129-157 + collate_fn :159-166
    gaussian_label(label, num_class, u, sig)           base_dataset.py:13-31 (host numpy, as in the reference)

`imgs_u8` [B, S, S, 3] uint8 BGR and `targets10` [nt, 10] = (image slot, class, x1, y1, ..., x4, y4 in pixels) are what the
reference holds at line 128; `flags[b]` bit 0 / 1 = the fliplr / flipud decisions (hyp['fliplr'], hyp['flipud'] draws, :133-138).
Returns (imgs [B, 3, S, S] fp32 RGB in [0, 1], targets [n, 7 | 187]) exactly as collate_fn hands them to train.py:183.

    BaseDataset                                        base_dataset.py:70-128: the sample composition itself — mosaic-4 / mosaic-9 /
                                                       mixup probabilities, load_image's resize + hsv, random_warping, letterbox —
                                                       for a WHOLE BATCH on the device (`assemble_batch`), the random draws made on the
                                                       host in the reference's order.  Subclasses (DOTA_dataset.py, UCASAOD_dataset.py)
                                                       only list files and parse labels, as in the reference.
Image decode (cv2.imread) stays on the host: `imread` is injectable; decoded uint8 images are kept in HBM by an ImagePool that is filled
on demand and bounded by `pool_budget_bytes` (LRU over slabs, pinned staging, thread-pool decode: datasets/augment.py).

    ImageDataset                                       base_dataset.py:59-81 (detect.py:12,43): a folder of images -> letterbox -> RGB ->
                                                       /255, batched on the device (`assemble_batch`, or `__getitems__` under a
                                                       torch DataLoader as detect.py:44 builds it).
    DeviceLoader                                       lib/load.py:19's DataLoader: batches assembled on a SIDE stream so that batch k + 1
                                                       is prepared under training step k.
"""
import os
import random as _py_random

import numpy as np
import torch

from .. import hip
from . import augment as A


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


def _default_imread(path):
    """cv2.imread when OpenCV is installed, else PIL (RGB -> BGR); the hot path never decodes — images are cached on the device."""
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        pass
    try:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    except ImportError:
        raise RuntimeError("ryolov4_amd.datasets: no image decoder (cv2 / PIL) — pass imread=callable(path) -> uint8 HWC BGR")


def _default_imsize(path):
    """(h, w) from the file header when PIL is there (no pixel decode); None = unknown until the image is decoded."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            w, h = im.size
        return h, w
    except Exception:
        return None


_POOL_CACHE = {}          # (image + label file lists, class list, device, budget, decoder) -> (ImagePool, parsed labels): test.py calls load_data once per evaluation


def clear_pool_cache():
    """Drop every shared pool (each pins at least one slab of device memory for as long as it is cached)."""
    _POOL_CACHE.clear()


class BaseDataset:
    """Same constructor and subclass contract as the reference's BaseDataset (datasets/base_dataset.py:70-77): subclasses fill
    `img_files` / `label_files` and implement `load_files(label_path) -> (polys float32 [n, 8], labels [n])`.  Differences by design:
    a sample is never built alone on a CPU worker — `assemble_batch(indices)` builds the whole batch on the device and returns what
    collate_fn returns; `__getitem__` / `collate_fn` exist for API parity and go through it.

    Random numbers: the reference draws from the global `random` and `numpy.random` modules; `rng=(random-like, numpy.random-like)`
    defaults to exactly those, and every draw is made in the reference's order (base_dataset.py:83-138, lib/augmentations.py:11,25,
    53-61), so the same seeds give the same sample composition (fixture G13: the imported reference's __getitem__ ran)."""

    def __init__(self, hyp, img_size, augment, csl, normalized_labels, device=None, imread=None, rng=None, imsize=None,
                 pool_budget_bytes=None, pool_slab_bytes=None, decode_workers=8, share_pool=True):
        self.hyp, self.img_size, self.augment, self.csl, self.normalized_labels = hyp, img_size, augment, csl, normalized_labels
        self.mosaic_border = [-img_size // 2, -img_size // 2]
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.imread = imread or _default_imread
        self.imsize = imsize or (_default_imsize if imread is None else (lambda path: None))
        self.rng = rng or (_py_random, np.random)
        self.img_files, self.label_files = [], []
        self._pool = self._labels = None
        self.pool_budget_bytes, self.pool_slab_bytes, self.decode_workers, self.share_pool = pool_budget_bytes, pool_slab_bytes, decode_workers, share_pool

    def shard(self, rank, world_size, pad=None):
        """Data parallel: this rank keeps every world_size-th file (and draws its mosaic partners among them), so the pool of a rank holds
        its shard only.  Call before the first batch.
        pad (default: self.augment, i.e. TRAINING datasets): every rank gets EXACTLY ceil(n / world) files (torch's DistributedSampler
        convention: the short shards wrap around to the head of the list): ranks then run the same number of batches of the same sizes — a
        rank with one batch more would sit in a gradient all-reduce that has no peer (n = 129, world 2, batch 64: 2 batches vs 1), and unequal
        last batches would be weighted equally by grad_scale = 1 / world.
        pad=False (EVALUATION datasets, augment=False: test.py:167-222 scores every image once): exact disjoint [rank::world] shards — a
        wrapped-around image would put its detections and its ground truth into the statistics twice and skew mAP; evaluation has no
        per-batch collective, so unequal shard lengths are harmless there."""
        if self._pool is not None:
            raise RuntimeError("BaseDataset.shard: call before the first batch is assembled")
        pad = self.augment if pad is None else pad
        n = len(self.img_files)
        if n:
            if pad:
                per = -(-n // world_size)
                idx = [(rank + k * world_size) % n for k in range(per)]
            else:
                idx = list(range(rank, n, world_size))
            self.img_files, self.label_files = [self.img_files[i] for i in idx], [self.label_files[i] for i in idx]
        return self

    def __len__(self):
        return len(self.img_files)

    def load_files(self, label_path):
        raise NotImplementedError

    # ------------------------------------------------------------------ cache: decoded images + parsed labels, resident
    def set_arrays(self, images, polys, labels):
        """Use already decoded images (uint8 HWC BGR, 1- or 3-channel) and parsed labels instead of reading files (the budget applies:
        with pool_budget_bytes the arrays are re-uploaded on demand like files are re-decoded)."""
        images = list(images)
        self._pool = A.ImagePool(count=len(images), decode=images.__getitem__, shapes=[tuple(np.asarray(im).shape[:2]) for im in images],
                                 device=self.device, budget_bytes=self.pool_budget_bytes, slab_bytes=self.pool_slab_bytes, workers=self.decode_workers)
        self._labels = {i: (np.asarray(p, dtype=np.float32).reshape(-1, 8), np.asarray(c, dtype=np.float32).reshape(-1))
                        for i, (p, c) in enumerate(zip(polys, labels))}
        if not self.img_files:
            self.img_files = [f"<array {i}>" for i in range(len(images))]
            self.label_files = list(self.img_files)

    def cache(self):
        """The pool of this dataset's images: nothing is decoded here — images are decoded (thread pool), staged in pinned memory and
        uploaded when a batch first needs them, and dropped slab-wise (LRU) once pool_budget_bytes is reached.  Image sizes come from the
        file headers (`imsize`) so that planning a batch does not need pixels.  Datasets over the same files on the same device share
        one pool (test.py calls load_data for every evaluation)."""
        if self._pool is None:
            files = self.img_files
            # the key holds the decoder OBJECT (a reference: an id() could be reused after garbage collection) and everything the parsed
            # labels depend on — label files, the class list of the concrete dataset, the label convention — so two datasets over the same
            # images with different classes or label directories never see each other's class indices
            key = (tuple(files), tuple(self.label_files), tuple(getattr(self, "category", None) or ()), type(self).__name__, bool(self.normalized_labels),
                   str(self.device), self.pool_budget_bytes, self.pool_slab_bytes, self.imread)
            if self.share_pool and key in _POOL_CACHE:
                self._pool, self._labels = _POOL_CACHE[key]
                return self._pool
            shapes = [self.imsize(p) for p in files]
            imread = self.imread
            self._pool = A.ImagePool(count=len(files), decode=lambda i: imread(files[i]), shapes=None if any(s is None for s in shapes) else shapes,
                                     device=self.device, budget_bytes=self.pool_budget_bytes, slab_bytes=self.pool_slab_bytes,
                                     workers=self.decode_workers)
            self._labels = {}
            if self.share_pool:
                _POOL_CACHE[key] = (self._pool, self._labels)
        return self._pool

    def labels_of(self, index):
        """(polys float32 [n, 8], classes float32 [n]) of image `index`, parsed on first use."""
        got = self._labels.get(index)
        if got is None:
            lp = self.label_files[index].rstrip()
            assert os.path.exists(lp), "Label file {} not found".format(lp)          # base_dataset.py:221
            p, c = self.load_files(lp)
            p = p.numpy() if isinstance(p, torch.Tensor) else p
            c = c.numpy() if isinstance(c, torch.Tensor) else (c if len(c) else np.zeros(0, np.float32))
            got = self._labels[index] = (np.asarray(p, dtype=np.float32).reshape(-1, 8), np.asarray(c, dtype=np.float32).reshape(-1))
        return got

    # ------------------------------------------------------------------ the draws, in the reference's order
    def _load_image_plan(self, index, items, luts):
        """load_image (base_dataset.py:170-186) as a table row: resized size, interpolation, the hsv tables of this use.
        Returns (row index into `items`, (h0, w0), (h, w))."""
        h0, w0 = self._pool.shapes[index]
        r = self.img_size / max(h0, w0)
        h, w, interp = h0, w0, A.INTERP_COPY
        if r != 1:
            interp = A.INTERP_AREA if (r < 1 and not self.augment) else A.INTERP_LINEAR
            w, h = int(w0 * r), int(h0 * r)
        lut = -1
        hy = self.hyp
        if self.augment and (hy["hsv_h"] or hy["hsv_s"] or hy["hsv_v"]):
            gains = self.rng[1].uniform(-1, 1, 3) * [hy["hsv_h"], hy["hsv_s"], hy["hsv_v"]] + 1
            lut = len(luts)
            luts.append(A.hsv_luts(gains))
        items.append((index, (h, w), interp, lut))
        return len(items) - 1, (h0, w0), (h, w)

    def _mosaic_plan(self, index, nine, items, luts):
        rnd, s, n = self.rng[0], self.img_size, len(self.img_files)
        if not nine:
            yc, xc = [int(rnd.uniform(-x, 2 * s + x)) for x in self.mosaic_border]
        indices = [index] + rnd.choices(range(n), k=8 if nine else 3)
        loaded = [self._load_image_plan(i, items, luts) for i in indices]
        if nine:
            yc, xc = [int(rnd.uniform(0, s)) for _ in self.mosaic_border]
        shapes = [hw for _, _, hw in loaded]
        uses = (A.mosaic9_uses if nine else A.mosaic4_uses)(shapes, [k for k, _, _ in loaded], s, yc, xc)
        return [(u, indices[j], loaded[j][1], loaded[j][2]) for j, u in enumerate(uses)]

    def _warp_draws(self, shape, border):
        rnd, hy = self.rng[0], self.hyp
        a = rnd.uniform(-hy["rotate"], hy["rotate"])
        sc = rnd.uniform(1 - hy["scale"], 1.1 + hy["scale"])
        tx = rnd.uniform(0.3 - hy["translate"], 0.3 + hy["translate"])
        ty = rnd.uniform(0.3 - hy["translate"], 0.3 + hy["translate"])
        return A.warp_matrix(shape, a, sc, tx, ty, border)

    # ------------------------------------------------------------------ one batch
    def assemble_batch(self, indices):
        """indices -> (paths, imgs [B, 3, S, S] fp32 RGB in [0, 1], targets [n, 7 | 187]) — what the reference's DataLoader hands to
        train.py:183 / test.py:183 after collate_fn, built on the device: ~10 launches per batch, one small host->device table each."""
        pool, s, dev = self.cache(), self.img_size, self.device
        rnd, nrd = self.rng
        B = len(indices)
        items, luts = [], []                       # resize + hsv stage: one row per source-image use
        canvases = []                              # per canvas: dict(kind, uses [(Use, dataset index, (h0, w0), (h, w))], M, slot)
        mixes, flags = [], np.zeros(B, dtype=np.uint8)
        out_of = []                                # per sample: index of the canvas that holds its image
        for slot, index in enumerate(indices):
            if self.augment and rnd.random() < self.hyp["mosaic"]:
                nine = not (rnd.random() < 0.8)
                cv = dict(kind="mosaic", uses=self._mosaic_plan(index, nine, items, luts), slot=slot)
                cv["M"], _ = self._warp_draws((2 * s, 2 * s), self.mosaic_border)
                canvases.append(cv)
                out_of.append(len(canvases) - 1)
                if nrd.random() < self.hyp["mixup"]:
                    nine2 = not (rnd.random() < 0.8)
                    other = rnd.randint(0, len(self.img_files) - 1)
                    cv2_ = dict(kind="mosaic", uses=self._mosaic_plan(other, nine2, items, luts), slot=slot)
                    cv2_["M"], _ = self._warp_draws((2 * s, 2 * s), self.mosaic_border)
                    canvases.append(cv2_)
                    mixes.append((out_of[-1], len(canvases) - 1, nrd.beta(8.0, 8.0)))
            else:
                k, hw0, hw = self._load_image_plan(index, items, luts)
                (nw, nh), (top, bottom, left, right), pad = A.pad_to_square_plan(hw, (s, s))
                cv = dict(kind="letterbox", item=k, hw0=hw0, hw=hw, new=(nh, nw), edges=(top, bottom, left, right), pad=pad, index=index, slot=slot,
                          M=None)
                if self.augment:
                    cv["M"], _ = self._warp_draws((top + nh + bottom, left + nw + right), (0, 0))
                canvases.append(cv)
                out_of.append(len(canvases) - 1)
            if self.augment and nrd.random() < self.hyp["fliplr"]:
                flags[slot] |= 1
            if self.augment and nrd.random() < self.hyp["flipud"]:
                flags[slot] |= 2
        # ---- pixels -----------------------------------------------------------------------------------------------------------------
        pool.ensure(it[0] for it in items)                        # decode + upload what is not resident (this batch's slabs are protected)
        stage, offs = A.resize_hsv_batch(pool, items, np.stack(luts) if luts else None)
        final = torch.empty((B, s, s, 3), dtype=torch.uint8, device=dev)
        mos = [c for c in canvases if c["kind"] == "mosaic"]
        warped = {}
        if mos:
            rects = []
            for ci, c in enumerate(mos):
                rects += [(offs[u.img], items[u.img][1][1], u.rect, ci) for u, _, _, _ in c["uses"]]
            big = A.paste(stage, rects, len(mos), 2 * s, 2 * s)
            small = A.warp_perspective(big, [c["M"] for c in mos], (s, s))
            for ci, c in enumerate(mos):
                warped[id(c)] = small[ci]
        for c in canvases:
            if c["kind"] == "letterbox":
                nh0, nw0 = c["hw"]
                src = stage[offs[c["item"]]:offs[c["item"]] + nh0 * nw0 * 3].view(nh0, nw0, 3)
                sq, _ = A.pad_to_square(src, (s, s))
                if tuple(sq.shape[:2]) != (s, s):
                    raise RuntimeError("assemble_batch: letterbox did not produce a square canvas")
                warped[id(c)] = A.warp_perspective(sq.unsqueeze(0), [c["M"]], (s, s))[0] if c["M"] is not None else sq
        mixed = {a: (b, r) for a, b, r in mixes}
        for slot in range(B):
            ci = out_of[slot]
            img = warped[id(canvases[ci])]
            if ci in mixed:
                b, r = mixed[ci]
                img = A.mixup(img, warped[id(canvases[b])], r)
            final[slot] = img
        # ---- labels: one table row per label of every source-image use, in the reference's concatenation order --------------------
        rows, mats = [], []
        for c in canvases:
            mat = -1
            if c["M"] is not None:
                mat = len(mats)
                mats.append(c["M"])
            if c["kind"] == "mosaic":
                for u, ds_index, hw0, hw in c["uses"]:
                    polys, cls = self.labels_of(ds_index)
                    rows.append(A.label_rows(polys, cls, c["slot"], hw0, hw, u, mat, self.normalized_labels))
            else:
                polys, cls = self.labels_of(c["index"])
                u = A.Use(c["item"], None, c["pad"], None, None)
                rows.append(A.label_rows(polys, cls, c["slot"], c["hw0"], c["hw"], u, mat, self.normalized_labels))
        rows = np.concatenate(rows) if rows else np.zeros(0, dtype=A.LABEL_ROW_DTYPE)
        # (mixup appends the second canvas' labels behind the first's: canvases of one sample are adjacent and in that order; samples are
        # in slot order, so the rows are already ordered the way collate_fn's torch.cat orders them)
        targets10 = A.label_stage(rows, np.stack(mats) if mats else None, dev)
        imgs, targets = finalize_batch(final, targets10, A._to_device(flags, dev), self.csl)
        return [self.img_files[i] for i in indices], imgs, targets

    # ------------------------------------------------------------------ API parity with torch.utils.data.Dataset users
    def __getitem__(self, index):
        paths, imgs, targets = self.assemble_batch([index])
        return paths[0], imgs[0], targets

    def collate_fn(self, batch):
        """base_dataset.py:159-166 for samples produced by __getitem__ (prefer assemble_batch: one pass for the whole batch)."""
        paths, imgs, targets = list(zip(*batch))
        for i, boxes in enumerate(targets):
            boxes[:, 0] = i
        return paths, torch.stack(imgs, 0), torch.cat(targets, 0)


class DeviceLoader:
    """Iterates a BaseDataset in batches assembled on the device (the role of torch.utils.data.DataLoader(dataset, batch_size, shuffle,
    num_workers=8, collate_fn=dataset.collate_fn) at lib/load.py:19): yields (paths, imgs, targets).

    Loader in the loop: the batch is assembled on a stream of the loader's own (`side_stream=True`, the default on a HIP device).  The
    consumer enqueues training step k asynchronously and asks for the next batch: its decode / planning runs on the host and its ~10
    launches run on the side stream WHILE step k occupies the compute stream; the compute stream only waits for the batch's event.
    rank / world_size: data parallel — the rank iterates (and pools) its shard of the files (BaseDataset.shard)."""

    def __init__(self, dataset, batch_size, shuffle, side_stream=True, rank=0, world_size=1, pad_shards=None):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        if world_size > 1:
            dataset.shard(rank, world_size, pad=pad_shards)      # None: padded for training (augment) datasets, exact for evaluation ones
        self._side = torch.cuda.Stream(device=dataset.device) if side_stream and dataset.device.type == "cuda" and torch.cuda.is_available() else None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def assemble(self, indices):
        if self._side is None:
            return self.dataset.assemble_batch(indices)
        main = torch.cuda.current_stream(self.dataset.device)
        with torch.cuda.stream(self._side):
            paths, imgs, targets = self.dataset.assemble_batch(indices)
            done = self._side.record_event()
        main.wait_event(done)
        imgs.record_stream(main)
        targets.record_stream(main)
        return paths, imgs, targets

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for a in range(0, n, self.batch_size):
            yield self.assemble(order[a:a + self.batch_size])


class ImageDataset:
    """The detect path's dataset (datasets/base_dataset.py:59-81; detect.py:12,43-44): every `*.ext` file of a folder, letterboxed to
    img_size x img_size on 114-grey (pad_to_square: cv2.resize INTER_LINEAR + copyMakeBorder), BGR -> RGB, float / 255.

    Same constructor, `__len__`, `__getitem__(index) -> (img_path, img [3, S, S])`.  The work is done per BATCH on the device:
    `assemble_batch(indices)` decodes on the host (imread injectable), uploads the uint8 images through pinned memory and runs three
    launches — ryolo_resize_hsv_batch (the resize of every image of the batch), ryolo_paste_rects (onto the grey canvases),
    ryolo_to_tensor.  `__getitems__` makes torch.utils.data.DataLoader(dataset, batch_size, shuffle=False) (detect.py:44) use that batch
    path; `loader(batch_size)` iterates it without the DataLoader's re-stacking copy."""

    def __init__(self, folder_path, img_size=416, ext="png", device=None, imread=None):
        import glob
        self.files = sorted(glob.glob(os.path.join(folder_path, "*.{}".format(ext))))
        self.img_size = img_size
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.imread = imread or _default_imread

    def __len__(self):
        return len(self.files)

    def assemble_batch(self, indices):
        """-> (paths, imgs [B, 3, S, S] fp32 RGB in [0, 1] on the device)."""
        s, dev = self.img_size, self.device
        paths = [self.files[i % len(self.files)] for i in indices]
        pool = A.ImagePool([np.asarray(self.imread(p)) for p in paths], dev)      # grey -> 3 channels inside the pool
        items, rects = [], []
        for k in range(len(paths)):
            (nw, nh), (top, bottom, left, right), _ = A.pad_to_square_plan(pool.shape(k), (s, s))
            if (top + nh + bottom, left + nw + right) != (s, s):
                raise RuntimeError("ImageDataset: letterbox did not produce a square canvas")
            items.append((k, (nh, nw), A.INTERP_COPY if (nh, nw) == pool.shape(k) else A.INTERP_LINEAR, -1))
            rects.append((nw, A.Placed(0, 0, left, top, nw, nh), k))
        stage, offs = A.resize_hsv_batch(pool, items)
        canv = A.paste(stage, [(offs[k], pitch, r, cv) for k, (pitch, r, cv) in enumerate(rects)], len(paths), s, s, fill=114)
        imgs = torch.empty((len(paths), 3, s, s), dtype=torch.float32, device=dev)
        hip.call("ryolo_to_tensor", hip.ptr(canv), len(paths), s, s, None, hip.ptr(imgs), hip.stream())
        return paths, imgs

    def __getitem__(self, index):
        paths, imgs = self.assemble_batch([index])
        return paths[0], imgs[0]

    def __getitems__(self, indices):
        paths, imgs = self.assemble_batch(list(indices))
        return [(p, imgs[k]) for k, p in enumerate(paths)]

    def loader(self, batch_size):
        for a in range(0, len(self.files), batch_size):
            yield self.assemble_batch(range(a, min(a + batch_size, len(self.files))))
