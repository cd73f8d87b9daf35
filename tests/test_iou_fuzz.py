"""Hardening the pin of SURVEY §8a N1-N3: the rotated-box pair function on >= 1e6 pairs, including the nasty
families (shared edges, angle differences below 1e-3 degrees, aspect ratios 1e-3 ... 1e3, coordinates up to 4096 * 16, near-duplicates,
concentric boxes, sub-pixel boxes), against an INDEPENDENT float64 computation (tests/iou_fuzz.py: Green's theorem over Cyrus-Beck
clipped edges — it shares nothing with the oracle's detectron2 scheme, nor with the scalar Sutherland-Hodgman clip of test_oracle_iou.py,
against which it is itself checked here).

What the fuzz found, and what is therefore asserted:
  * away from degenerate configurations the restated detectron2 algorithm (fp32, intersection points + Graham hull) is within 1e-5 of
    the exact IoU — the threshold flips that decide NMS keep sets (iou > 0.2 / 0.65) are not at risk from arithmetic error;
  * the algorithm has a GROSS failure mode: when vertices of one box lie (nearly) on edges of the other, near-duplicate intersection
    points make its polar-angle sort order by rounding noise and the hull loses (or doubles) area — IoU 1/3 or 2.3 for boxes whose true
    IoU is 0.999999.  Rates measured here: ~3e-3 of near-duplicate pairs (relative perturbation 2e-6), ~1e-5 of the other families.
    detectron2's own unit tests (test_iou_issue_2154 / _2167) cover EXACTLY identical boxes only.  This is the upstream algorithm's
    behaviour as published (oracle/rotated_iou.c restates it statement by statement; detectron2 itself is absent and cannot confirm).
    WHAT BACKS "UPSTREAM BEHAVIOUR" (round 5): the failure is a property of the SCHEME (points ordered by the sign of a
    cross product that is rounding noise for near-coincident points), not of one sort routine.  detectron2 orders the points in two ways —
    the CUDA path by an O(n^2) exchange sort with the predicate `cross < -1e-6 || (|cross| < 1e-6 && dist_i > dist_j)`, the CPU path by
    std::sort with `|cross| < 1e-6 ? |A|^2 < |B|^2 : cross > 0` — and BOTH are restated here: the exchange sort in oracle/rotated_iou.c (the
    contract: lib/general.py:177 runs nms_rotated on GPU tensors), the CPU path as a second build of the same file with the REAL std::sort of
    this image's libstdc++ (oracle/hull_stdsort.cpp).  Measured by test_both_upstream_hull_orderings_fail_on_near_duplicates below
    (profiles/r05_iou_sort_variants.json, 200 000 near-duplicate pairs + 40 000 of every other family): exchange sort 3.1e-3 gross failures
    on near-duplicates, std::sort 2.5e-3 — 466 pairs fail under BOTH, 154 only under the exchange sort, 34 only under std::sort — and on
    every other family the two orderings give bit-identical IoUs (0 gross failures in 7 x 40 000 pairs).  So neither ordering is "the correct one" that the other deviates from; the GPU contract is the exchange sort.
    The build does NOT "fix" it: the parity contract of §8a N1 is the keep set of detectron2's algorithm, so
  * the HIP pair function must be BIT-IDENTICAL to the oracle on every one of the 1.2 M + 1 M pairs, failures included.
The deviation histograms (GPU vs float64) are written to gpurun_out/r05_iou_fuzz.json and committed under profiles/."""
import json
import os

import numpy as np
import pytest

import oracle
from tests import iou_fuzz as F
from tests.test_oracle_iou import KNOWN, sh_iou

REGULAR = ("general", "shared_edges", "tiny_angle_difference", "class_offset_coordinates", "concentric")
GROSS = 1e-2                                        # a deviation beyond this is a hull failure, not rounding


def test_fp64_reference_against_scalar_clip_and_known_answers():
    for b1, b2, expect in KNOWN:
        assert abs(F.iou_fp64(np.array([b1]), np.array([b2]))[0] - expect) < 1e-6
    for name, (a, b) in F.families(400, seed=11).items():
        g = F.iou_fp64(a, b)
        s = np.array([sh_iou([float(v) for v in x], [float(v) for v in y]) for x, y in zip(a, b)])
        assert np.abs(g - s).max() < 1e-8, name


def _check_family(name, dev, out):
    h = F.histogram(dev)
    gross = float((dev > GROSS).mean())
    h["gross_failure_rate"] = gross
    out[name] = h
    ok = dev[dev <= GROSS]
    if name in REGULAR:
        assert gross < 1e-4 and ok.max() < 1e-4 and np.quantile(dev, 0.999) < 1e-5, (name, h)
    elif name == "near_identical":
        assert gross < 2e-2 and np.quantile(dev, 0.99) < 1e-3, (name, h)             # the failure family: rate reported, bounded
    else:                                                                             # extreme aspect ratios, sub-pixel boxes: fp32 geometry
        assert gross < 5e-4 and np.quantile(dev, 0.999) < 1e-3, (name, h)


def test_oracle_against_fp64_on_all_families_cpu():
    """The oracle alone (no GPU): 8 x 40 000 pairs."""
    out = {}
    for name, (a, b) in F.families(40000, seed=21).items():
        _check_family(name, np.abs(oracle.diag_iou_rotated(a, b).astype(np.float64) - F.iou_fp64(a, b)), out)


def test_both_upstream_hull_orderings_fail_on_near_duplicates():
    """detectron2's CUDA-path exchange sort (the contract) and its CPU-path std::sort ordering (oracle/hull_stdsort.cpp, the real
    std::sort) on the fuzz families: gross-failure rates (|IoU - fp64 IoU| > 1e-2) of both, how many pairs fail under both / only one, and
    that away from the failure family the two agree to rounding.  Writes gpurun_out/r05_iou_sort_variants.json (committed under profiles/)."""
    rep = {"what": "gross failures (|IoU - independent fp64 IoU| > 1e-2) of the two hull orderings detectron2 ships: CUDA path = O(n^2) exchange "
                   "sort (oracle/rotated_iou.c, the parity contract), CPU path = std::sort with the distance-tie comparator (oracle/hull_stdsort.cpp, "
                   "libstdc++ of this image)", "gross_threshold": GROSS, "families": {}}
    for name, (a, b) in F.families(40000, seed=41).items():
        if name == "near_identical":
            a, b = F.families(200000, seed=42)["near_identical"]
        ref = F.iou_fp64(a, b)
        cu = oracle.diag_iou_rotated(a, b, hull_sort="cuda").astype(np.float64)
        cp = oracle.diag_iou_rotated(a, b, hull_sort="cpu").astype(np.float64)
        fcu, fcp = np.abs(cu - ref) > GROSS, np.abs(cp - ref) > GROSS
        both_ok = ~fcu & ~fcp
        rep["families"][name] = {"pairs": int(len(a)), "gross_rate_exchange_sort": float(fcu.mean()), "gross_rate_std_sort": float(fcp.mean()),
                                 "fail_under_both": int((fcu & fcp).sum()), "only_exchange_sort": int((fcu & ~fcp).sum()),
                                 "only_std_sort": int((~fcu & fcp).sum()),
                                 "max_abs_difference_where_both_are_right": float(np.abs(cu - cp)[both_ok].max()),
                                 "bit_identical_fraction": float((cu == cp).mean())}
        if name == "near_identical":
            # the claim of the docstring: BOTH orderings have the failure mode, at the same order of magnitude
            assert 5e-4 < fcu.mean() < 2e-2 and 5e-4 < fcp.mean() < 2e-2, rep["families"][name]
            assert 0.2 < fcp.mean() / fcu.mean() < 5.0
        else:
            assert fcu.mean() < 5e-4 and fcp.mean() < 5e-4, (name, rep["families"][name])
        if name != "near_identical":                                        # (there, partial hull losses below the gross line exist too)
            assert np.abs(cu - cp)[both_ok].max() < 1e-4, name              # where neither fails they are the same number up to rounding
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open(os.path.join("gpurun_out", "r05_iou_sort_variants.json"), "w"), indent=1)


@pytest.mark.gpu
def test_hip_pair_function_on_two_million_pairs():
    import torch
    from ryolov4_amd.lib import general
    from ryolov4_amd.synth import synth_nms_boxes
    dev = torch.device("cuda:0")
    report = {"what": "ryolo_diag_iou_rotated / ryolo_box_iou_rotated (csrc/rotated_iou.h) vs an independent float64 IoU (tests/iou_fuzz.py); "
                      "|deviation| histograms per family; bit_identical_to_oracle = every pair equals oracle/rotated_iou.c bit for bit",
              "families": {}, "gross_threshold": GROSS}
    total = 0
    for name, (a, b) in F.families(150000, seed=31).items():
        got = general.diag_iou_rotated(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
        ref = oracle.diag_iou_rotated(a, b)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, int((got != ref).sum()))
        _check_family(name, np.abs(got.astype(np.float64) - F.iou_fp64(a, b)), report["families"])
        report["families"][name]["bit_identical_to_oracle"] = True
        total += len(a)
    # the [N, M] entry point (test.py:135) on a clustered set: 1024 x 1024 pairs, most of them disjoint or near-duplicates of each other
    bx, _ = synth_nms_boxes(2048, "C", seed=9)
    m = general.pairwise_iou_rotated(torch.from_numpy(bx[:1024]).to(dev), torch.from_numpy(bx[1024:]).to(dev)).cpu().numpy()
    ia, ib = np.repeat(np.arange(1024), 1024), np.tile(np.arange(1024), 1024) + 1024
    ref = oracle.diag_iou_rotated(bx[ia], bx[ib]).reshape(1024, 1024)
    assert np.array_equal(m.view(np.uint32), ref.view(np.uint32))
    d = np.abs(m.reshape(-1).astype(np.float64) - F.iou_fp64(bx[ia], bx[ib]))
    report["pairwise_1024x1024_clustered"] = dict(F.histogram(d), gross_failure_rate=float((d > GROSS).mean()), bit_identical_to_oracle=True,
                                                  overlapping_pairs=int((ref > 0).sum()))
    assert (d > GROSS).mean() < 1e-4 and d[d <= GROSS].max() < 1e-4
    total += m.size
    report["pairs"] = total
    assert total >= 2_000_000
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open(os.path.join("gpurun_out", "r05_iou_fuzz.json"), "w"), indent=1)
