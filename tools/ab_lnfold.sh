#!/bin/bash
# A/B of the LayerNorm-folded frames against the LayerNorm-kernel frames, same box, interleaved (run on the GPU box).
for rep in 1 2; do
  for flag in "" "--no-ln-fold"; do
    python bench.py --model L --batch 8 --template-size 256 --search-size 384 --steps 40 --warmup 10 --blocks 5 --no-cpu-baseline --no-batched $flag 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('L b8  %-13s %8.1f fps  %.3f ms' % ('$flag' or 'ln-fold', d['value'], d['ms_per_step']))"
    python bench.py --batch 32 --steps 40 --warmup 10 --blocks 5 --no-cpu-baseline --no-batched $flag 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B b32 %-13s %8.1f fps  %.3f ms' % ('$flag' or 'ln-fold', d['value'], d['ms_per_step']))"
    python bench.py --batch 8 --steps 100 --warmup 10 --blocks 5 --no-cpu-baseline --no-batched $flag 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B b8  %-13s %8.1f fps  %.3f ms' % ('$flag' or 'ln-fold', d['value'], d['ms_per_step']))"
  done
done
