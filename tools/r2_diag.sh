#!/bin/bash
# round-2 first lease: reproduce the driver's cold-box hang of the two-rank bench under torch.distributed.run, then the suite
mkdir -p gpurun_out/r2a
export T2N_BENCH_BACKEND=gloo T2N_BENCH_SAME_DEVICE=1 OMP_NUM_THREADS=4 T2N_BENCH_DEADLINE_S=150
( time timeout 260 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --train-iters 2 --train-warmup 1 ) > gpurun_out/r2a/torchrun.out 2> gpurun_out/r2a/torchrun.err
echo "torchrun rc $?" >> gpurun_out/r2a/torchrun.out
unset T2N_BENCH_BACKEND T2N_BENCH_SAME_DEVICE OMP_NUM_THREADS T2N_BENCH_DEADLINE_S
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r2a/gpu_tests.log 2>&1
tail -5 gpurun_out/r2a/gpu_tests.log
tail -3 gpurun_out/r2a/torchrun.out
