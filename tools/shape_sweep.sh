#!/bin/bash
# bench.py over a few non-default shapes (each must complete; prints ms/step)
for a in "--dim 768" "--tgt 64" "--tgt 300" "--batch 4" "--rate 12" "--rate 2" "--frames 3000 --batch 16" "--heads 16" "--layers 6 --dim 512"; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a > gpurun_out/sweep.log 2>&1 || { echo "FAILED: $a"; tail -5 gpurun_out/sweep.log; exit 1; }
  echo "$a: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep.log)"
done
