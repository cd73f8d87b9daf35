"""bench.py host logic that needs no GPU: the committed PMC profile must resolve for every conv kernel class of the default workload
(kernel names carry template arguments that change when a kernel gains one — the roofline's `traffic` went null that way once)."""
import argparse

import bench


def test_committed_pmc_profile_resolves_for_the_default_workload():
    """The committed PMC profile either belongs to THESE sources (then every conv kernel class of the default workload resolves in it) or
    is reported stale and contributes nothing — old bytes are never attached to new kernels."""
    import json
    import os
    args = argparse.Namespace(ver="yolov7", mode="kfiou", size=800, nc=16, batch=64)
    path = os.path.join(bench.ROOT, bench.PMC_PROFILE)
    fresh = os.path.exists(path) and json.load(open(path)).get("source_sha256") == bench.source_sha256()
    t = bench.pmc_traffic(args)
    if fresh:
        for cls in ("conv_gemm_kernel<128x128,1x1>", "conv_gemm_kernel<128x128>", "conv_gemm_kernel<256x64>", "conv3x3_patch_kernel<256x128>", "conv3x3_patch_kernel<256x64>",
                    "conv_wgrad_kernel<128>", "conv3x3_wgrad_kernel<128x9x32>"):
            assert t.get(cls, 0) > 1_000_000, (cls, t.get(cls))
        assert 150e9 < bench.pmc_step_bytes(args) < 400e9
    else:
        assert t == {} and bench.pmc_step_bytes(args) == 0
    other = argparse.Namespace(ver="yolov7", mode="kfiou", size=800, nc=16, batch=8)
    assert bench.pmc_traffic(other) == {} and bench.pmc_step_bytes(other) == 0
