#!/bin/bash
# Vector / scalar / LDS instruction counts of the fused chain kernel with parts of the proposal switched off (diagnostic
# switches GSM_PROPOSE_DBG: 32 no coefficient items, 2 no stage-1 MFMA loop, 4 no stage-2 loop, 8 no field emit):
# differences = instructions of each part.   scripts/pmc_ablate.sh   -> gpurun_out/ablate/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/ablate; mkdir -p $out; export TMPDIR=/tmp; cd $root
for dbg in 0 32 34 38 46; do
  GSM_PROPOSE_DBG=$dbg rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $out/d$dbg -- python3 scripts/pmc_fused.py 64 > $out/d$dbg.log 2>&1
  python3 - <<EOF
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/d$dbg/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("dbg $dbg per chain-step:", {k: round(sum(v) / len(v) / (1024 * 64), 1) for k, v in sorted(acc.items())})
EOF
done | tee $out/summary.txt
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete
