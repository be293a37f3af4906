#!/bin/bash
# Run on the GPU box (via gpurun): SQ issue / stall counters of the three hot kernels, a few per pass
# (counters only: no trace domains mixed with --pmc).
set -e
OUT=gpurun_out/pmc_deep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1 || echo "pass $i failed: $grp"
done <<'GRP'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_CYCLES SQ_BUSY_CU_CYCLES
GRP
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_deep/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        if "spx_fwd" in k or "spx_bwd" in k or "spx_bank_bwd" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_deep/summary.txt", "w") as o:
    for k, v in agg.items():
        o.write(k + "\n")
        for c in sorted(v):
            o.write("  %-34s %16.0f\n" % (c, sum(v[c]) / len(v[c])))
print(open("gpurun_out/pmc_deep/summary.txt").read())
PY
