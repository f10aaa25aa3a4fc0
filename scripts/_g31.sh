set -e
mkdir -p gpurun_out
bash scripts/pmc_kernel.sh sgs_ sgs256d -- scripts/sgs_bench.py --chains 256 --iters 100 --philox > gpurun_out/pmc_sgs256d.out 2>&1
grep -E "sgs_weights_kernel +(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES)" gpurun_out/pmc_sgs256d/summary.txt
bash scripts/sgs_profile.sh > gpurun_out/r3h_sgs_profile.out 2>&1 || { tail -20 gpurun_out/r3h_sgs_profile.out; exit 1; }
tail -11 gpurun_out/r3h_sgs_profile.out
python scripts/sgs_bench.py --chains 4 --iters 640 --pcg64 2>&1 | grep small-scale
python scripts/sgs_bench.py --chains 64 --iters 640 --pcg64 2>&1 | grep small-scale
python scripts/sgs_bench.py --chains 64 --iters 640 --philox 2>&1 | grep small-scale
