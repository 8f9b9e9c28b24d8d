#!/bin/bash
# gpurun call 3 of round 3: what a chunk of the P3 GEMM is made of -- knock-outs and SQ counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
P=$ROOT/tools/scratch/gemm3_probe
{
for cfg in 0 1 3; do
  for k in 0 1 2 4 3 5 6 7; do $P single 4096 4096 1024 3 1 $cfg 10 $k; done
done
for k in 0 1 2 4 6; do $P single 1024 2048 1024 3 4 0 20 $k; done
for k in 0 1 2 4 6; do $P single 2048 1024 1024 0 4 0 20 $k; done
} > $OUT/r03_g3_knock.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in 0 1; do
 for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/g3pmc_${cfg}_$tag -o p -- $P single 4096 4096 1024 3 1 $cfg 3 0 > $OUT/g3pmc_${cfg}_$tag.log 2>&1
  python - <<PY > $OUT/r03_g3_pmc_${cfg}_$tag.txt
import csv, glob, collections
rows=[]
for f in glob.glob("$OUT/g3pmc_${cfg}_$tag/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:<36} n={len(vals):3d} mean={sum(vals)/len(vals):16.1f}")
PY
  rm -rf $OUT/g3pmc_${cfg}_$tag
 done
done
cat $OUT/r03_g3_knock.txt
