#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "randomised" > gpurun_out/fuzz.log 2>&1; echo "rc $?" >> gpurun_out/fuzz.log
