set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/s4b; mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1"
for k in -1 0 3 4 5; do
  if [ $k = -1 ]; then unset LM_SPONGE_WGS_PER_CU; else export LM_SPONGE_WGS_PER_CU=$k; fi
  $B > $OUT/b_$k.json 2> $OUT/b_$k.err
  python - $OUT/b_$k.json $k <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("sponge wgs/cu", sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "sponge ms", round(d["roofline_sponge"]["ms_per_step"],3), "hot", round(d["hot_path"]["ms_per_step"],3))
P
done | tee $OUT/sponge_sweep.txt
unset LM_SPONGE_WGS_PER_CU
LM_STAGE_TIMES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --no-whole-node > $OUT/stage.json 2> $OUT/stage_times.txt
tail -150 $OUT/stage_times.txt | head -150
