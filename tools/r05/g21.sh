#!/bin/bash
mkdir -p gpurun_out/r05t
timeout 1200 python -m pytest tests/test_gpu_temporal_block.py -x -q -m gpu 2>&1 | tail -15
