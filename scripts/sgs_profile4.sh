# small-scale chain at BASELINE configs[0]'s literal size (4 chains): rocprofv3 kernel stats, pcg64 and Philox modes
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r04}
L=gpurun_out/${T}_sgs4.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_sgsprof4 -o p4 -- python scripts/sgs_bench.py --chains 4 --iters 300 --philox > $L 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_sgsprof4p -o p4p -- python scripts/sgs_bench.py --chains 4 --iters 300 --pcg64 >> $L 2>&1
grep "small-scale" $L
