#!/bin/bash
# The HOST side of an N-GPU node on a one-GPU box: bench.py under torch.distributed.run with every rank on GPU 0 (LM_BENCH_SINGLE_DEVICE=1,
# gloo).  The GPU is shared, so step times mean nothing; what is read is host_busy / cpu per rank-step (bench.py: "ranks").
# usage: tools/ranks_on_one_gpu.sh <out file> [wait modes="auto spin"] [rank counts="2 4 8"]
OUT=${1:-gpurun_out/ranks_on_one_gpu.txt}
MODES=${2:-"auto spin"}
RANKS=${3:-"2 4 8"}
cd "$(dirname "$0")/.."
echo "bench.py under torch.distributed.run, LM_BENCH_SINGLE_DEVICE=1 (every rank on GPU 0: the GPU is shared, the HOST side is what an N-GPU node would see); $(nproc) CPUs" > $OUT
for mode in $MODES; do
  for n in $RANKS; do
    LM_WAIT_MODE=$mode LM_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 10 --warmup 2 --scale-log 3 --no-cpu-baseline --no-whole-node --dist-backend gloo 2> /tmp/ranks_$n.err | tail -1 > /tmp/ranks_$n.json
    python - "$mode" "$n" >> $OUT <<'PY'
import json, sys
mode, n = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"/tmp/ranks_{n}.json").read())
    r = d["ranks"]["per_rank"]
    hb = [x["host_busy_ms_per_step"] for x in r]; cpu = [x["cpu_ms_per_step"] for x in r]
    print(f"LM_WAIT_MODE={mode}: {n} ranks on ONE GPU (--scale-log 3, {d['ranks']['cpus_available']} CPUs): step {d['ms_per_step']:.2f} ms, value {d['value']:.0f} sigs/s; "
          f"host_busy per rank {min(hb):.2f}-{max(hb):.2f} ms, CPU time per rank and step {min(cpu):.2f}-{max(cpu):.2f} ms, "
          f"jitter between ranks mean {d['ranks']['step_jitter_ms']['mean']:.3f} / max {d['ranks']['step_jitter_ms']['max']:.3f} ms")
except Exception as e:
    print(f"LM_WAIT_MODE={mode}: {n} ranks: FAILED {e!r}", open(f"/tmp/ranks_{n}.err").read()[-600:])
PY
  done
done
cat $OUT
