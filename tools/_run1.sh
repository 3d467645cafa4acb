set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/s4a; mkdir -p $OUT
cd $ROOT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --no-whole-node > $OUT/bench_rocprof.json 2> $OUT/trace.err
cd $ROOT
python tools/launch_list.py $OUT/trace > $OUT/launch_list.txt
python tools/timeline.py $OUT/trace > $OUT/timeline.txt
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT/trace -name "*.csv" ! -name "*kernel_stats*" -delete
tail -c 600 $OUT/bench_base.json
