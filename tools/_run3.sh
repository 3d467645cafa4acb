set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/s4c; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_commit_gpu.py tests/test_whir_gpu.py tests/test_golden_r03.py tests/test_golden.py -x -q -m gpu > $OUT/tests1.log 2>&1
tail -5 $OUT/tests1.log
timeout 900 python -m pytest tests/test_execution_gpu.py -x -q -m gpu -k "not full_size and not recursion" > $OUT/tests2.log 2>&1
tail -5 $OUT/tests2.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"],3), "hot", round(d["hot_path"]["ms_per_step"],3), "exch", d["exchanges"]["per_step"])
print({k:round(v,3) for k,v in d["stages_ms"].items()})
P
LM_MERKLE_NO_MID=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1 > $OUT/bench_nomid.json 2> $OUT/bench_nomid.err
python - $OUT/bench_nomid.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("NO_MID ms_per_step", round(d["ms_per_step"],3), "hot", round(d["hot_path"]["ms_per_step"],3))
print({k:round(v,3) for k,v in d["stages_ms"].items()})
P
