#!/bin/bash
# Where the fused chain kernel's cycles go, from the SQ cycle counters (one rocprofv3 --pmc pass per group).
#   scripts/pmc_cycles.sh [tag]   -> gpurun_out/cycles_<tag>/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-x}; out=$root/gpurun_out/cycles_$tag; mkdir -p $out; export TMPDIR=/tmp; cd $root
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ"
 "SQC_DCACHE_HITS SQC_DCACHE_MISSES TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --output-format csv -d $out/g$i -- python3 scripts/pmc_fused.py 64 > $out/g$i.log 2>&1 || echo "group $i failed: $g"
  python3 - <<EOF
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/g$i/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-36s %16.0f per launch  %12.1f per chain-step" % (k, sum(v) / len(v), sum(v) / len(v) / (1024 * 64)))
EOF
  i=$((i+1))
done | tee $out/summary.txt
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete
