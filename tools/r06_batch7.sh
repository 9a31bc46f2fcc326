#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_i_pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_i_bench.json 2>gpurun_out/r06_i_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 --bench-a-steps 0 --fp32-steps 0 > gpurun_out/r06_i_bench2.json 2>/dev/null
