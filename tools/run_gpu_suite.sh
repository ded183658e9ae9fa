#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; echo "rc $?" >> gpurun_out/suite.log
