#!/bin/bash
# rocprofv3 counter passes (one group per pass, counters only) over a workload, summarised for the kernels whose name contains $1:
#   scripts/pmc_kernel.sh <kernel substring> <out tag> -- <python script and args>
# Output: gpurun_out/pmc_<tag>/summary.txt (per-dispatch averages of every counter).
sub=$1; tag=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/pmc_$tag; mkdir -p $out; export TMPDIR=/tmp; cd $root
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 "$@" > $out/g$i.log 2>&1; echo "group $i ($grp) rc=$?"
done
python3 - "$out" "$sub" <<'PY' | tee $out/summary.txt
import csv, glob, collections, sys
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, k), v in sorted(acc.items()):
    print(f"{name:40s} {k:30s} dispatches {len(v):5d}  mean {sum(v) / len(v):.6g}  total {sum(v):.6g}")
PY
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
