import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch
dev = torch.device("cuda:0")
for B in (8, 64):
    net = Yolo(16, CFG, "kfiou", "yolov7"); net.apply(bench.weights_init_normal); net.to(dev).train()
    crit = ComputeKFIoULoss(net, HYP)
    imgs, tg = synth_batch(B, 800, 16, False, seed=1); imgs, tg = imgs.to(dev), tg.to(dev)
    rt = net.runtime(dev)
    def step():
        outs = net(imgs, training=True)
        loss, _ = crit(outs, tg, sync_items=False)
        loss.backward()
        rt.sgd_step(1e-3)
    for _ in range(5): step()
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"B={B}: host enqueue of one step {min(hs)*1e3:.2f} ms (median {sorted(hs)[5]*1e3:.2f}), one step alone incl. sync {min(ts)*1e3:.2f} ms, pipelined {(time.perf_counter()-t0)/20*1e3:.2f} ms/step; launches fwd {len(rt._graphs[(B,800,800,True,False)].fwd)} bwd {len(rt._graphs[(B,800,800,True,False)].bwd)}")
    del net, crit
