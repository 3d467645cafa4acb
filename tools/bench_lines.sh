#!/bin/bash
# The plain bench lines of a round (BASELINE configs and side measurements) -> gpurun_out/$1/.  Run AFTER the counter summaries
# of the same sources have been published to profiles/ (tools/publish_profiles.sh): bench.py then reports traffic / alu from them.
set -u
TAG=${1:-r06_final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --no-cpu-baseline --inflight 1 --verify --equal-oracle > $OUT/bench_verify.json 2> $OUT/bench_verify.err
python bench.py --no-cpu-baseline --inflight 1 --log-inv-rate 2 --verify --equal-oracle > $OUT/bench_config2_rate4.json 2> /dev/null
python bench.py --no-cpu-baseline --inflight 1 --soundness capacity --verify --equal-oracle > $OUT/bench_capacity.json 2> /dev/null
python bench.py --no-cpu-baseline --inflight 1 --soundness capacity --log-inv-rate 2 --verify --equal-oracle > $OUT/bench_capacity_rate4.json 2> /dev/null
python bench.py --shape recursion --log-inv-rate 2 --inflight 1 --steps 5 --verify --equal-oracle --profile-all > $OUT/bench_recursion_shape.json 2> $OUT/bench_recursion_shape_kernels.txt
for c in 2 4 6 8 10 12; do python bench.py --no-cpu-baseline --inflight $c --steps 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['inflight']['proofs_in_flight'], round(d['inflight']['value']), round(d['inflight']['ms_per_proof'],2), round(d['ms_per_step'],2))"; done > $OUT/inflight_sweep.txt
# BASELINE configs[3] as far as the recursion program is assembled: four genuine 775-signature leaves at rate 1/4, then the root step
python bench.py --shape whir-recursion --log-inv-rate 2 --steps 5 --warmup 2 --equal-oracle > $OUT/bench_whir_recursion.json 2> $OUT/bench_whir_recursion.err
