"""Microbenchmark of the BatchNorm + activation backward passes (ryolo_bn_act_bwd: reduce -> finalize -> apply) through the C ABI on the
step's large layer shapes; statistics-only calls (dy1 = null) time reduce + finalize alone.  usage: python tools/bench_bnact.py [reps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
hip.lib(); S.check_layouts()
st = hip.stream()
SHAPES = [(64 * 400 * 400, 64, 64), (64 * 400 * 400, 64, 256), (64 * 200 * 200, 128, 128), (64 * 200 * 200, 256, 256), (64 * 100 * 100, 256, 256),
          (64 * 100 * 100, 512, 512), (64 * 50 * 50, 1024, 1024), (64 * 25 * 25, 1024, 1024), (8 * 100 * 100, 256, 256),
          (8 * 50 * 50, 256, 256), (8 * 50 * 50, 128, 128), (8 * 25 * 25, 256, 256), (8 * 25 * 25, 512, 512), (8 * 25 * 25, 1024, 1024)]      # (the 8-image step's small maps)
print("lib", hip.LIB_PATH)
gen = torch.Generator(device=dev).manual_seed(0)
for M, Cc, ld in SHAPES:
    y = torch.randn(M, ld, device=dev, generator=gen).to(torch.bfloat16)
    dz = (torch.randn(M, ld, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    dy = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
    co = torch.zeros(4, Cc, device=dev)
    co[1] = 1.0; co[2] = 1.0 + 0.1 * torch.randn(Cc, device=dev, generator=gen); co[3] = 0.1 * torch.randn(Cc, device=dev, generator=gen)
    bco = torch.zeros(3, Cc, device=dev)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    nblk, rpb = S.I(), S.I()
    hip.call("ryolo_bn_act_bwd_blocks", M, Cc, nblk, rpb)
    part = torch.zeros(nblk.value + 64, 2, Cc, device=dev)
    res = {}
    for mode in ("stats", "full"):
        p = S.BnActParams()
        p.y1, p.ld1, p.co1 = y.data_ptr(), ld, co.data_ptr()
        p.M, p.C, p.act = M, Cc, 3
        p.dz, p.lddz = dz.data_ptr(), ld
        if mode == "full":
            p.dy1, p.lddy1 = dy.data_ptr(), ld
        p.partial = part.data_ptr()
        for _ in range(2): hip.call("ryolo_bn_act_bwd", p, dg.data_ptr(), db.data_ptr(), None, None, bco.data_ptr(), 0, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): hip.call("ryolo_bn_act_bwd", p, dg.data_ptr(), db.data_ptr(), None, None, bco.data_ptr(), 0, st)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / reps * 1e3
    z = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
    p = S.BnActParams()
    p.y1, p.ld1, p.co1 = y.data_ptr(), ld, co.data_ptr()
    p.z, p.ldz, p.M, p.C, p.act = z.data_ptr(), ld, M, Cc, 3
    for _ in range(2): hip.call("ryolo_bn_act_fwd", p, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): hip.call("ryolo_bn_act_fwd", p, st)
    e1.record(); torch.cuda.synchronize()
    res["fwd"] = e0.elapsed_time(e1) / reps * 1e3
    del z
    by = M * Cc * 2
    print(f"M{M:9d} C{Cc:5d} ld{ld:5d}: fwd {res['fwd']:8.1f} us {2 * by / res['fwd'] / 1e6:6.2f} TB/s | reduce+fin {res['stats']:8.1f} us {2 * by / res['stats'] / 1e6:6.2f} TB/s | apply {res['full'] - res['stats']:8.1f} us "
          f"{3 * by / (res['full'] - res['stats']) / 1e6:6.2f} TB/s | total {res['full']:8.1f} us  sums {float(bco.abs().sum()):.6e} dg {float(dg.abs().sum()):.6e}")
    del y, dz, dy
