#!/bin/bash
# All profiler passes of a round on the GPU box (rocprofv3; counters in their own passes, never combined with traces):
#   scripts/profile_round.sh <tag>      ->  gpurun_out/prof_<tag>/...
# 1 kernel trace + stats of the default bench workload; 2 FETCH_SIZE, 3 WRITE_SIZE and 4.. SQ counter groups over
# scripts/pmc_fused.py (calibration stream copy + three 2048-step fused launches = the bench launch length).
set -u
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_profiled.json 2> $out/trace.err
echo "trace rc=$?"
python3 scripts/pmc_fused.py 2048 > $out/workload.jsonl 2> $out/workload.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 scripts/pmc_fused.py 2048 > $out/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 scripts/pmc_fused.py 2048 > $out/write.log 2>&1; echo "write rc=$?"
i=0
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/sq$i -- python3 scripts/pmc_fused.py 2048 > $out/sq$i.log 2>&1; echo "sq$i rc=$?"
done
python3 scripts/make_pmc_traffic.py $out/fetch $out/write $out/workload.jsonl $out/pmc_traffic.json $out/sq1 $out/sq2 $out/sq3 $out/sq4 > $out/pmc_summary.log 2>&1; echo "summary rc=$?"
python3 scripts/summarize_pmc.py $out/pmc_counters.csv $out/fetch $out/write $out/sq1 $out/sq2 $out/sq3 $out/sq4 > /dev/null 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# keep the merged output small: drop the raw per-dispatch databases / csv
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
tail -3 $out/bench_profiled.json | cut -c1-300
head -5 $out/kernel_stats.csv
tail -25 $out/pmc_summary.log
