set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/s4d; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gkr_gpu.py -x -q -m gpu > $OUT/tests_gkr.log 2>&1
tail -3 $OUT/tests_gkr.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1"
summ() { python - $1 "$2" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "hot", round(d["hot_path"]["ms_per_step"],3), "exch", d["exchanges"]["per_step"], {k.split(":")[0][:12]:round(v,3) for k,v in d["stages_ms"].items()})
P
}
for rep in 1 2; do
  LM_GKR_TAIL_W=16 $B > $OUT/b_w16_$rep.json 2>/dev/null; summ $OUT/b_w16_$rep.json "W16 rep$rep"
  LM_GKR_TAIL_W=64 $B > $OUT/b_w64_$rep.json 2>/dev/null; summ $OUT/b_w64_$rep.json "W64 rep$rep"
  LM_GKR_TAIL_W=32 $B > $OUT/b_w32_$rep.json 2>/dev/null; summ $OUT/b_w32_$rep.json "W32 rep$rep"
  LM_MERKLE_NO_MID=1 $B > $OUT/b_nomid_$rep.json 2>/dev/null; summ $OUT/b_nomid_$rep.json "W64 NOMID rep$rep"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --no-whole-node > $OUT/bench_rocprof.json 2> $OUT/trace.err
cd $ROOT
python tools/launch_list.py $OUT/trace > $OUT/launch_list.txt
python tools/timeline.py $OUT/trace > $OUT/timeline.txt
find $OUT/trace -name "*.csv" ! -name "*kernel_stats*" -delete
head -3 $OUT/timeline.txt
