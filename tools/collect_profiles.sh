#!/bin/bash
# Round profile set (run on the GPU box through gpurun).  Outputs under gpurun_out/$1/:
#   stats_inflight1/  rocprofv3 --kernel-trace --stats of one prover alone (kernel durations without contention)
#   stats_default/    the same for the default bench command (6 proofs in flight)
#   pmc_fetch/, pmc_write/   FETCH_SIZE / WRITE_SIZE passes (separate, as the MI355X guide prescribes), one prover
#   bench_default.json, bench_verify.json   plain bench lines
set -u
TAG=${1:-r01_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B1="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1"
B3="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_inflight1 -- $B1 > $OUT/bench_inflight1_under_rocprof.json 2> $OUT/stats1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- $B3 > $OUT/bench_default_under_rocprof.json 2> $OUT/stats3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_valu -- $B1 > /dev/null 2> $OUT/pmc_valu.err
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify > $OUT/bench_verify.json 2> $OUT/bench_verify.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > $OUT/bench_inflight1.json 2> $OUT/bench_inflight1.err
python bench.py --no-cpu-baseline --log-inv-rate 2 > $OUT/bench_config3_rate4.json 2> /dev/null
python bench.py --no-cpu-baseline --log-inv-rate 2 --inflight 1 --steps 3 > $OUT/bench_config3_rate4_inflight1.json 2> /dev/null
python bench.py --no-cpu-baseline --host-resident > $OUT/bench_host_resident.json 2> /dev/null
python bench.py --no-cpu-baseline --host-resident --inflight 1 --steps 3 > $OUT/bench_host_resident_inflight1.json 2> /dev/null
python bench.py --shape recursion --log-inv-rate 2 --inflight 1 --steps 3 --verify --profile-all > $OUT/bench_recursion_shape_inflight1.json 2> $OUT/bench_recursion_shape_kernels.txt
python bench.py --shape recursion --log-inv-rate 2 --steps 3 > $OUT/bench_recursion_shape.json 2> /dev/null
for c in 1 2 4 6 8 10 12; do python bench.py --no-cpu-baseline --inflight $c --steps 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['proofs_in_flight_per_gpu'], round(d['value']), round(d['ms_per_step'],2))"; done > $OUT/inflight_sweep.txt
python tools/pmc_summary.py $OUT 5 $OUT/pmc_bench.json > /dev/null
python tools/valu_summary.py $OUT 5 $OUT/valu_bench.json > $OUT/valu_bench.txt
python tools/launch_seq.py $OUT/stats_inflight1 k_air_round 5 60 > $OUT/air_round_launches.txt
for k in k_gkr_fold_round k_air_round k_fold_round; do python tools/launch_hist.py $OUT/stats_inflight1 $k 5; done > $OUT/launch_hist.txt
(nproc; lscpu | grep "Model name") > $OUT/host.txt
# keep the summaries, drop the raw per-launch CSVs of the counter passes (tens of MB)
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_valu
find $OUT -name "*_kernel_trace.csv" -delete
ls $OUT
