#!/bin/bash
# Vector / scalar instruction counts and wave cycles of chain_strip_kernel with parts switched off (GSM_PROPOSE_DBG: 32 no
# coefficient items, 2 no stage-1 MFMA loop, 4 no stage-2 loop, 8 no field emit, 128 no phase D, 256 no phase A,
# 512 no state loads, 1024 no candidate read / commit): differences = instructions of each part.  Results of the chains are
# wrong with any switch set.   scripts/pmc_ablate_strip.sh [steps]  -> gpurun_out/ablate_strip/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/ablate_strip; mkdir -p $out; export TMPDIR=/tmp; cd $root
steps=${1:-256}
for dbg in ${ABL_LIST:-0 128 384 896 1920 1952 1954 1958 1966}; do
  GSM_PROPOSE_DBG=$dbg rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/d$dbg -- python3 scripts/pmc_fused.py $steps > $out/d$dbg.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/d$dbg/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_strip_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("dbg $dbg per chain-step:", {k: round(sum(v) / len(v) / (1024 * $steps), 1) for k, v in sorted(acc.items())})
PY
done | tee $out/summary.txt
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete
