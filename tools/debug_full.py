import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from oracle import ref_model, ref_ops
from tests.bf16_emu import emulate_bf16
from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.lib.loss import ComputeKFIoULoss, ComputeCSLLoss
DEV="cuda:0"
def rel(a,b):
    a,b=a.detach().double().flatten(),b.detach().double().flatten(); return float((a-b).norm()/(b.norm()+1e-12))
for ver in ("yolov7","yolov4","yolov5"):
  for mode in ("kfiou","csl"):
    nc=2
    net=Yolo(nc,CFG,mode,ver); sd=fill_state(net.state_dict()); net.load_state_dict(sd); net.to(DEV)
    orc=ref_model.Yolo(nc,CFG,mode,ver); orc.load_state_dict(sd)
    na=3 if mode=="csl" else 18
    def five(b):
        B_,_,gs,_=b.shape; return b.view(B_,na,-1,gs,gs).permute(0,1,3,4,2)
    x=torch.rand(2,3,64,64,generator=torch.Generator().manual_seed(1))
    net.eval(); orc.eval()
    with torch.no_grad():
        outs,inf=net(x.to(DEV),training=False); hm=orc.head_maps(x)
    print(ver,mode,"eval64 fp32-oracle", [round(rel(a.cpu(),five(b)),4) for a,b in zip(outs,hm)])
    emulate_bf16(orc)
    with torch.no_grad():
        hm=orc.head_maps(x)
    print(ver,mode,"eval64 emu-oracle ", [round(rel(a.cpu(),five(b)),4) for a,b in zip(outs,hm)])
    x=torch.rand(2,3,160,160,generator=torch.Generator().manual_seed(2))
    net.train(); orc.train()
    tg=synth_targets(2,6,nc,mode=="csl",seed=3,img_size=160)
    t0=time.time(); outs_o=orc(x,True); loss_o,items_o=ref_ops.compute_loss(outs_o,tg,orc.anchors,nc,mode,HYP); loss_o.backward()
    crit=(ComputeCSLLoss if mode=="csl" else ComputeKFIoULoss)(net,HYP)
    outs=net(x.to(DEV),training=True)
    print(ver,mode,"train160 emu-oracle", [round(rel(a.cpu(),b),4) for a,b in zip(outs,outs_o)])
    loss,items=crit(outs,tg.to(DEV)); loss.backward()
    print("   loss", items["total_loss"], float(items_o["total_loss"]))
    gp=torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double(); go=torch.cat([q.grad.flatten() for q in orc.parameters()]).double()
    print("   grad cos", float((gp@go)/(gp.norm()*go.norm())), "norm ratio", float(gp.norm()/go.norm()))
    errs=sorted(((rel(p.grad.cpu(),q.grad),n) for (n,p),(_,q) in zip(net.named_parameters(),orc.named_parameters())),reverse=True)
    print("   worst", [(round(e,3),n) for e,n in errs[:4]], "median", round(errs[len(errs)//2][0],3))
