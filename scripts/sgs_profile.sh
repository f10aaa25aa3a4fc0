# small-scale chain: wall-clock rates (scripts/sgs_bench.py) and rocprofv3 kernel stats of the driver configuration
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r04}
L=gpurun_out/${T}_sgsb.log
python scripts/sgs_bench.py --chains 4 --iters 400 --cpu-iters 10 > $L 2>&1
python scripts/sgs_bench.py --chains 4 --iters 2000 --philox >> $L 2>&1
python scripts/sgs_bench.py --chains 4 --iters 2000 --pcg64 >> $L 2>&1
python scripts/sgs_bench.py --chains 256 --iters 200 --philox >> $L 2>&1
python scripts/sgs_bench.py --chains 4 --iters 1000 --philox --light --no-transform >> $L 2>&1
python scripts/sgs_bench.py --chains 256 --iters 400 --philox --light --no-transform >> $L 2>&1
python scripts/sgs_bench.py --chains 4 --iters 1000 --light >> $L 2>&1
python scripts/sgs_bench.py --grid 256 --chains 16 --iters 300 --philox >> $L 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_sgsprof4 -o p4 -- python scripts/sgs_bench.py --chains 4 --iters 300 --philox >> $L 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_sgsprof256 -o p256 -- python scripts/sgs_bench.py --chains 256 --iters 100 --philox >> $L 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_sgsprofL -o pL -- python scripts/sgs_bench.py --chains 256 --iters 200 --philox --light --no-transform >> $L 2>&1
grep "small-scale" $L
