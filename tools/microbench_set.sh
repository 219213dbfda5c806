#!/bin/bash
# GPU-box aid: the 3x3x3 shapes of the 96^3 B=2 joint step, one line each (bf16, lazy input)
for a in "2 8 8 96" "2 16 8 96" "2 16 16 48" "2 32 16 48" "2 32 32 24" "2 64 32 24" "2 64 64 12" "2 128 64 12"; do
  python tools/microbench_conv.py $a 50 bf16 lazy
done
