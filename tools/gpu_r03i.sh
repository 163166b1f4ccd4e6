#!/bin/bash
mkdir -p gpurun_out/r03i
#timeout 600 python bench.py --train --steps 2 --warmup 1 2>gpurun_out/r03i/train.err | tail -1 > gpurun_out/r03i/r03i_train_256x512.json; cat gpurun_out/r03i/r03i_train_256x512.json; tail -3 gpurun_out/r03i/train.err
timeout 900 python bench.py --train --steps 1 --warmup 1 --train-height 576 --train-width 1024 2>gpurun_out/r03i/train_full.err | tail -1 > gpurun_out/r03i/r03i_train_576x1024.json; cat gpurun_out/r03i/r03i_train_576x1024.json; tail -3 gpurun_out/r03i/train_full.err
