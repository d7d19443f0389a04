#!/bin/bash
# interleaved A/B of two library builds on one box: tools/probes/ab/lib_a.so vs lib_b.so (bench.py default workload, extra flags passed through)
for r in 0 1 2; do
  for v in a b; do
    python tools/probes/bench_lib.py tools/probes/ab/lib_$v.so --steps 100 --warmup 20 --no-cpu-baseline --no-batched "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $r lib_$v %.1f frames/s %.4f ms' % (d['value'], d['ms_per_step']))"
  done
done
