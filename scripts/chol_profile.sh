# configs[3] (512x512 grid x 1024 chains, precomputed-Cholesky proposals): bench line + rocprofv3 kernel stats
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3_chol; mkdir -p $out
python bench.py --grid 512 --chains 1024 --generator cholesky --classes 2 --batch 128 --inner 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o chol -- python3 bench.py --grid 512 --chains 1024 --generator cholesky --classes 2 --batch 128 --inner 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_profiled.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete; find $out -name "*.db" -delete
tail -1 $out/bench_line.json | cut -c1-1500
head -14 $out/kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
