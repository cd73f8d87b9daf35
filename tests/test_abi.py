"""CPU test: libryolo_hip.so loads and exports every symbol include/ryolo.h declares (no compute without a GPU),
and the product path refuses CPU tensors instead of silently falling back."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ryolo.h")).read()
    return sorted(set(re.findall(r"\b(ryolo_[a-z0-9_]+)\s*\(", txt)) - {"ryolo_stream_t"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from ryolov4_amd import hip
    L = hip.lib()
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), name
    assert set(hip.exported_symbols()) == set(declared), set(hip.exported_symbols()) ^ set(declared)


def test_product_path_has_no_cpu_fallback():
    from ryolov4_amd.lib import general
    with pytest.raises(RuntimeError):
        general.post_process(torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError):
        general.nms_rotated(torch.zeros(3, 5), torch.zeros(3), 0.5)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "r-yolov4_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(d, f)
                assert "liboracle" not in src, os.path.join(d, f)
