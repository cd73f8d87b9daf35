#!/bin/bash
# PMC passes for the single-layer weight-gradient microbenchmark (run on the GPU box from the repo root): bash tools/pmc_wgrad.sh <tag> <bench_wgrad args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
ARGS="$@"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ TCC_HIT TCC_MISS TCC_REQ" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  CHECK=0 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_${TAG}_$tag -o p -- python $R/tools/bench_wgrad.py $ARGS > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(R+"/gpurun_out/pmcw_${TAG}_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "wgrad" not in k or "reduce" in k: continue
        agg[r["Counter_Name"]]["v"]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("== ${TAG}: $ARGS")
for k,v in sorted(agg.items()):
    print(f"{k:36s} per-dispatch {v['v']/cnt[k]:16.1f}  (n={cnt[k]})")
PY
rm -rf $R/gpurun_out/pmcw_${TAG}_*
