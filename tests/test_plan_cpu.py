"""CPU tests of the host logic of the static-graph engine: plans for every ver x mode build without a GPU (buffers on
CPU, nothing launched), the module tree keeps the reference's state_dict ABI, and gradient-write resolution is sane."""
import pytest
import torch

from oracle import ref_model
from ryolov4_amd.engine.runtime import Runtime
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG


@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
@pytest.mark.parametrize("mode", ["csl", "kfiou"])
def test_state_dict_abi_matches_reference_layout(ver, mode):
    ours = Yolo(2, CFG, mode, ver).state_dict()
    ref = ref_model.Yolo(2, CFG, mode, ver).state_dict()          # pinned against the imported reference by make_golden.py
    assert list(ours.keys()) == list(ref.keys())
    for k in ours:
        assert ours[k].shape == ref[k].shape, k


@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
@pytest.mark.parametrize("mode,nc", [("csl", 2), ("kfiou", 16)])
@pytest.mark.parametrize("training", [True, False])
def test_plan_builds(ver, mode, nc, training):
    m = Yolo(nc, CFG, mode, ver)
    m.train(training)
    rt = Runtime(m, torch.device("cpu"))
    g = rt.graph(2, 64, 64, training)
    na = 3 if mode == "csl" else 18
    attrs = nc + (185 if mode == "csl" else 6)
    assert [tuple(h["out"].shape) for h in g.heads] == [(2, na, 8, 8, attrs), (2, na, 4, 4, attrs), (2, na, 2, 2, attrs)]
    nconv = sum(1 for x in m.modules() if isinstance(x, torch.nn.Conv2d))
    names = [n for _, _, n in g.fwd]
    # 3x3 stride-1 stems (v4, v7) bypass im2col + GEMM; training plans run the direct kernel twice (statistics only, then with
    # BatchNorm + activation fused: the raw output is never stored) and have ONE fused backward launch for the layer
    direct_stem = 0 if ver == "yolov5" else 1
    assert names.count("ryolo_stem3x3_fwd") == direct_stem * (2 if training else 1)
    # eval plans re-parameterise every RepConv (3x3 + 1x1 -> one 3x3 GEMM); training plans keep both branches
    from ryolov4_amd.model.blocks import C3, CSP, ELAN1, ELAN2, SPPCSPC, RepConv
    nrep = sum(1 for x in m.modules() if isinstance(x, RepConv))
    assert nrep == (3 if ver == "yolov7" else 0)
    # cv1 + cv2 of these blocks read the same input: ONE GEMM launch (training plans: every such block; eval plans: only where the two
    # outputs are adjacent concat slices, i.e. the ELAN blocks — the folded-BN epilogue writes one destination)
    npair = sum(1 for x in m.modules() if isinstance(x, (ELAN1, ELAN2) if not training else (ELAN1, ELAN2, CSP, C3, SPPCSPC)))
    assert npair > 0 or (not training and ver != "yolov7")
    assert names.count("ryolo_conv_gemm") == nconv - direct_stem - (0 if training else nrep) - npair
    if not training:
        assert [n for _, _, n in g.wprep].count("ryolo_repconv_fold") == nrep
    if training:
        bnames = [n for _, _, n in g.bwd]
        assert bnames.count("ryolo_conv_wgrad") + bnames.count("ryolo_stem3x3_bwd") == nconv - npair
        assert bnames.count("ryolo_stem3x3_bwd") == direct_stem and bnames.count("ryolo_stem3x3_wgrad") == 0
        assert bnames.count("ryolo_conv_gemm") == nconv - 1 - npair  # every conv but the stem has a data gradient (shared by siblings)
        assert names.count("ryolo_bn_finalize_slice") == 2 * npair
        for p in m.parameters():
            assert rt.grad_view(p).shape == p.shape
    else:
        assert g.bwd == []


def test_unknown_mode_raises_like_reference():
    with pytest.raises(NotImplementedError):
        Yolo(2, CFG, "smooth_l1", "yolov4")
    with pytest.raises(RuntimeError):
        Yolo(2, CFG, "csl", "yolov4")(torch.zeros(1, 3, 64, 64), True)          # CPU tensor: no fallback


@pytest.mark.parametrize("ver", ["yolov4", "yolov7"])
def test_pretrained_first_552_entries_rule(ver):
    """train.py:74-86 verbatim on this build's Yolo: a reference checkpoint (here: the oracle's state_dict, whose keys / shapes /
    order are pinned to the imported reference) is cut to its first 552 entries, merged into model.state_dict() and loaded
    strictly — the checkpoint ABI is the reference's."""
    from oracle import ref_model
    from ryolov4_amd.synth import fill_state
    pretrained = fill_state(ref_model.Yolo(16, CFG, "kfiou", ver).state_dict())
    model = Yolo(16, CFG, "kfiou", ver)
    pretrained_dict = {k: v for i, (k, v) in enumerate(pretrained.items()) if i < 552}
    model_dict = model.state_dict()
    model_dict.update(pretrained_dict)
    model.load_state_dict(model_dict)                              # strict
    got = model.state_dict()
    for i, (k, v) in enumerate(pretrained.items()):
        if i < 552:
            assert torch.equal(got[k], v), k
    assert len(pretrained) >= 552


@pytest.mark.parametrize("training", [True, False])
def test_stream_plan_of_yolov7(training):
    """Host-side bookkeeping of the multi-stream schedule (Graph.run): one forked forward branch per ELAN1 / ELAN2 / MaxConv block on
    lane 1, the two early detection heads on lane 2, a join point on a main-stream entry after every lane-1 branch, and (training)
    every regular weight gradient on the weight-gradient stream."""
    from ryolov4_amd.model.blocks import ELAN1, ELAN2, MaxConv
    m = Yolo(2, CFG, "kfiou", "yolov7")
    m.train(training)
    g = Runtime(m, torch.device("cpu")).graph(2, 64, 64, training)
    nblocks = sum(1 for x in m.modules() if isinstance(x, MaxConv))        # (the ELAN siblings share one GEMM launch: nothing to fork)
    firsts = [i for i, (first, lane) in g.fwd_side.items() if first]
    assert sum(1 for i in firsts if g.fwd_side[i][1] == 1) == nblocks
    assert sum(1 for i in firsts if g.fwd_side[i][1] == 2) == 2
    assert len(g.fwd_join) == nblocks
    for j in g.fwd_join:
        k = j
        while k in g.fwd_side:                      # a join lands on the next main-stream entry (Graph.run defers it)
            k += 1
        assert k < len(g.fwd)
    # forked entries are contiguous runs starting with a `first` entry
    for i in sorted(g.fwd_side):
        first, lane = g.fwd_side[i]
        assert first or ((i - 1) in g.fwd_side and g.fwd_side[i - 1][1] == lane)
    if training:
        names = [n for _, _, n in g.bwd]
        # side stream: the weight gradients and, behind the three detection heads' ones, the pass that finishes their ImplicitM chain rule
        assert g.side_idx and all(names[i] in ("ryolo_conv_wgrad", "ryolo_head_wgrad_finish") for i in g.side_idx)
        assert names.count("ryolo_head_wgrad_finish") in (0, 3)
        assert len(g.side_idx) == names.count("ryolo_conv_wgrad") + names.count("ryolo_head_wgrad_finish")   # yolov7's stem runs the direct kernel: every conv_wgrad is regular
    else:
        assert not g.side_idx


@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
def test_main_stream_wgrad_never_shares_the_split_k_workspace_with_the_side_stream(ver):
    """Backward runs on two streams; launches of one stream are ordered, launches of different streams are not: a weight gradient
    that stays on the main stream (yolov5's im2col stem) must not use the split-K slabs of the side-stream weight gradients."""
    from ryolov4_amd.engine import structs as S
    m = Yolo(2, CFG, "kfiou", ver)
    m.train()
    g = Runtime(m, torch.device("cpu")).graph(2, 64, 64, True)
    side_ws, main_ws = set(), set()
    for i, (fn, args, name) in enumerate(g.bwd):
        if name != "ryolo_conv_wgrad":
            continue
        p = args[0]._obj
        assert isinstance(p, S.WgradParams) and p.partial
        (side_ws if i in g.side_idx else main_ws).add(p.partial)
    assert side_ws and not (side_ws & main_ws)
    if ver == "yolov5":
        assert main_ws                                   # the 6x6 stride-2 stem goes through im2col + the generic wgrad on the main stream


@pytest.mark.parametrize("ver", ["yolov7", "yolov4", "yolov5"])
@pytest.mark.parametrize("training", [True, False])
def test_arena_layout_is_conflict_free_and_small(ver, training):
    """engine/arena.py: no two buffers whose lifetimes intersect share bytes; training plans need about half of the one-tensor-per-buffer
    sum (the gradient twins ride in dead activations), inference plans a fifth; both planning passes emit the same tapes."""
    from ryolov4_amd.engine import arena
    m = Yolo(16, CFG, "kfiou", ver)
    rt = Runtime(m, torch.device("cpu"))
    g = rt.graph(2, 128, 128, training)
    lay = g.layout
    assert lay is not None and arena.check(lay)
    assert lay.total <= (0.62 if training else 0.35) * lay.sum_bytes
    assert g.arena.numel() == lay.total
    rt2 = Runtime(m, torch.device("cpu"))
    rt2.buffer_reuse = False
    g2 = rt2.graph(2, 128, 128, training)
    assert g2.layout is None and [e[2] for e in g2.fwd] == [e[2] for e in g.fwd] and [e[2] for e in g2.bwd] == [e[2] for e in g.bwd]
    # a weight gradient's operands stay put until the main stream has waited for it: lifetimes of side-stream reads end at a later
    # weight gradient's position, never before their own
    F = len(g.fwd)
    for k, (lo, hi) in lay.life.items():
        assert 0 <= lo <= hi < F + len(g.bwd)
