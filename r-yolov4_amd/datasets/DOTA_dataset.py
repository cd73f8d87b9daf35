"""DOTA label files for the device loader (reference: datasets/DOTA_dataset.py:9-49): `images/*.png` with `annfiles/*.txt`, one object
per line: eight polygon coordinates, the category name (blanks written as '-'), then anything else (difficulty), blank separated."""
import glob
import os

import numpy as np
import torch

from .base_dataset import BaseDataset


def parse_polygon_lines(lines, sep, first, name_at, category):
    """Label text -> (polys float32 [n, 8], labels int64 [n]): fields split on `sep`, coordinates at fields first ... first + 7, the
    category name at field `name_at` (KeyError for a name that is not in `category`, as the reference's dict lookup)."""
    polys, labels = [], []
    for line in lines:
        f = line.split(sep)
        polys.append([float(f[first + k]) for k in range(8)])
        labels.append(category[f[name_at]])
    if not labels:
        return torch.zeros((0, 8), dtype=torch.float32), []             # the reference returns the empty python list here too
    return torch.tensor(np.asarray(polys, dtype=np.float64)).type(torch.float32), torch.tensor(labels)


class DOTADataset(BaseDataset):
    def __init__(self, data_dir, class_names, hyp, augment, img_size, csl, normalized_labels=False, **device_kw):
        super().__init__(hyp, img_size, augment, csl, normalized_labels, **device_kw)
        self.img_files = sorted(glob.glob(os.path.join(data_dir, "images", "*.png")))
        self.label_files = [p.replace("images", "annfiles").replace(".png", ".txt") for p in self.img_files]
        self.category = {name.replace(" ", "-"): i for i, name in enumerate(class_names)}

    def load_files(self, label_path):
        with open(label_path, "r") as fh:
            return parse_polygon_lines(fh.readlines(), " ", 0, 8, self.category)
