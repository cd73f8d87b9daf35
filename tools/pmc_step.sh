#!/bin/bash
# HBM traffic of every kernel class of one training step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (MI355X_MICROARCH.md: they do not fit one pass; never combined with sys/hip tracing).  Run on the GPU box from the repo root:
#   bash tools/pmc_step.sh [round tag, default r02]   ->  gpurun_out/<tag>_pmc_step_traffic.json  (copy to profiles/)
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STEPS=3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_$c -o p -- python $R/bench.py --steps $STEPS --warmup 1 --no-kernel-timing --no-cpu-baseline --no-loader --no-b8 --no-infer > $R/gpurun_out/pmc_step_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json, os, re
R=os.environ["GRAFT_REPO_ROOT"]
raw={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob(R+f"/gpurun_out/pmc_step_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]!=c: continue
            k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
            agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    for k in agg: raw.setdefault(k,{})[c]=(agg[k],cnt[k])
kern={}
for k,v in raw.items():
    f,n=v.get("FETCH_SIZE",(0.0,0)); w,n2=v.get("WRITE_SIZE",(0.0,0)); n=max(n,n2)
    if not n: continue
    kern[k]={"launches":n,"fetch_kib_raw":round(f/n,1),"write_kib":round(w/n,1),"hbm_bytes_per_launch":int((2*f+w)*1024/n)}
steps=kern.get("sgd_nesterov_kernel",{}).get("launches",0)
doc={"_doc":"HBM traffic per launch from rocprofv3 PMC passes over bench.py (batch 64, yolov7 kfiou nc=16 800^2; launches of sgd_nesterov_kernel = steps_profiled, warm-up and loss read-back steps included): FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes (tools/pmc_step.sh); units KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so hbm_bytes = (2*FETCH + WRITE)*1024 (MI355X_MICROARCH.md, HBM section; the x2 applies to wide 16-B/lane reads, kernels with 4-byte gathers are over-counted by it)",
     "steps_profiled":steps,"kernels":kern}
import hashlib
h=hashlib.sha256()
for f in sorted(glob.glob(R+"/r-yolov4_amd/csrc/*.hip")+glob.glob(R+"/r-yolov4_amd/csrc/*.h")+glob.glob(R+"/include/*.h")+glob.glob(R+"/r-yolov4_amd/engine/*.py")+glob.glob(R+"/r-yolov4_amd/model/*.py")):
    h.update(open(f,"rb").read())
doc["source_sha256"]=h.hexdigest()      # bench.py reports traffic only while the kernels and the plan are the ones that were profiled
json.dump(doc,open(R+f"/gpurun_out/${TAG}_pmc_step_traffic.json","w"),indent=1)
skip=("nms_","bitonic_","compose_kernel","topk_")
tot=sum(v["hbm_bytes_per_launch"]*v["launches"] for k,v in kern.items() if not k.startswith(skip))/max(steps,1)
print("steps",steps,"GB per step",round(tot/1e9,1))
for k,v in sorted(kern.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:28]:
    print(k[:70].ljust(70), v["launches"], round(v["hbm_bytes_per_launch"]*v["launches"]/max(steps,1)/1e9,2),"GB/step")
PY
