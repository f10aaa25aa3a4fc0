# sgs_weights_kernel: vector / scalar instructions per simulated cell of library builds (rocprofv3 --pmc, 256 chains x 100 iterations, Philox mode)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in hip "$@"; do
  export GSM_LIB=$GRAFT_REPO_ROOT/mcmc_gpu_amd/libgsm_$v.so
  rm -rf gpurun_out/pmc_wv
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_wv -- python3 scripts/sgs_bench.py --chains 256 --iters 100 --philox --cells > gpurun_out/pmc_wv.log 2>&1
  python3 - "$v" <<PY
import csv, glob, collections, re, sys
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_wv/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgs_weights" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
cells = int(re.search(r"iterations: (\d+)", open("gpurun_out/pmc_wv.log").read()).group(1))
print(f"{sys.argv[1]}: simulated cells {cells}; per cell: VALU {acc['SQ_INSTS_VALU'] / cells:.0f}, SALU {acc['SQ_INSTS_SALU'] / cells:.0f}")
PY
done
rm -rf gpurun_out/pmc_wv
