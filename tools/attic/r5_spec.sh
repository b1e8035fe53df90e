#!/bin/bash
# speculative (device-rows) train step: tests + timing beside the counted step, same box
mkdir -p gpurun_out/r5_spec
timeout 600 python -m pytest tests/test_train_step.py -x -q -m gpu > gpurun_out/r5_spec/tests.txt 2>&1
tail -5 gpurun_out/r5_spec/tests.txt
timeout 600 python tools/experiments/spec_step_timing.py > gpurun_out/r5_spec/timing.txt 2>&1
cat gpurun_out/r5_spec/timing.txt | tail -20
