#!/bin/bash
# HBM traffic of every kernel class of one training step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (MI355X_MICROARCH.md: they do not fit one pass; never combined with sys/hip tracing).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline > $R/gpurun_out/pmc_step_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json, os, re
R=os.environ["GRAFT_REPO_ROOT"]
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob(R+f"/gpurun_out/pmc_step_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]!=c: continue
            k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
            agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    for k in agg: out.setdefault(k,{})[c]={"sum":agg[k],"n":cnt[k]}
json.dump(out,open(R+"/gpurun_out/pmc_step.json","w"),indent=1)
for k,v in sorted(out.items(), key=lambda kv:-kv[1].get("FETCH_SIZE",{}).get("sum",0))[:25]:
    print(k[:60].ljust(60), {c:(round(v[c]["sum"]/v[c]["n"],1), v[c]["n"]) for c in v})
PY
