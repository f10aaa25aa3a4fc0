# sgs_sequence_kernel: average duration (rocprofv3 kernel stats, 4 chains, Philox mode) of library builds with parts of the kernel left out
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in hip "$@"; do
  export GSM_LIB=$GRAFT_REPO_ROOT/mcmc_gpu_amd/libgsm_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/seq_abl_$v -o p -- python scripts/sgs_bench.py --chains 4 --iters 300 --philox > gpurun_out/seq_abl_$v.log 2>&1 || true
  echo "$v: $(grep sgs_sequence_kernel gpurun_out/seq_abl_$v/p_kernel_stats.csv | cut -d, -f2-4,6,7)  weights: $(grep sgs_weights_kernel gpurun_out/seq_abl_$v/p_kernel_stats.csv | cut -d, -f4)"
done
