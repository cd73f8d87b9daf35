"""UCAS-AOD label files for the device loader (reference: datasets/UCASAOD_dataset.py:11-51): `*.png` with `*.txt` beside it, one object
per line, TAB separated: the category name, then eight polygon coordinates (further fields ignored)."""
import glob
import os

from .base_dataset import BaseDataset
from .DOTA_dataset import parse_polygon_lines


class UCASAODDataset(BaseDataset):
    def __init__(self, data_dir, class_names, hyp, augment, img_size, csl, normalized_labels=False, **device_kw):
        super().__init__(hyp, img_size, augment, csl, normalized_labels, **device_kw)
        self.img_files = sorted(glob.glob(os.path.join(data_dir, "*.png")))
        self.label_files = [p.replace(".png", ".txt") for p in self.img_files]
        self.category = {name.replace(" ", "-"): i for i, name in enumerate(class_names)}

    def load_files(self, label_path):
        with open(label_path, "r") as fh:
            return parse_polygon_lines(fh.readlines(), "\t", 1, 0, self.category)
