#!/bin/bash
# LDS bank-conflict share of one conv launch: bash tools/pmc_lds.sh <bench_conv.py args>   (EPI from the environment)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_lds -o p -- python $R/tools/bench_conv.py "$@" > /tmp/pmc_lds.log 2>&1
grep "TF/s" /tmp/pmc_lds.log
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("/tmp/pmc_lds/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_gemm" not in r["Kernel_Name"] and "conv3x3" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
d={k:agg[k]/cnt[k] for k in agg}
print({k:round(v/1e6,2) for k,v in d.items()}, "conflict share %.3f"%(d.get("SQ_LDS_BANK_CONFLICT",0)/max(d.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
