#!/bin/bash
# Round profile set (run on the GPU box through gpurun): kernel stats, FETCH/WRITE PMC passes (separate, as the
# MI355X guide prescribes), and the default bench line.  Outputs under gpurun_out/$1/.
set -u
TAG=${1:-r01_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify > $OUT/bench_verify.json 2> $OUT/bench_verify.err
ls -R $OUT | head -40
