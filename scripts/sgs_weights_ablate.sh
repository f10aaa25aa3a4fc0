# sgs_weights_kernel: average duration (rocprofv3 kernel stats, Philox mode) of library builds that return after each part of the kernel
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=${CHAINS:-4}
for v in hip "$@"; do
  export GSM_LIB=$GRAFT_REPO_ROOT/mcmc_gpu_amd/libgsm_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w_abl_$v -o p -- python scripts/sgs_bench.py --chains $C --iters 200 --philox > gpurun_out/w_abl_$v.log 2>&1 || true
  echo "$v ($C chains): $(grep sgs_weights_kernel gpurun_out/w_abl_$v/p_kernel_stats.csv | cut -d, -f2-4,6,7)"
done
