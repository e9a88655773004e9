#!/bin/bash
mkdir -p gpurun_out/r05t
{
for k in 0 1 2 3 4 5; do python tools/stream_map_probe.py $k 2>&1 | grep dummy; done
for q in 2 3 5 6; do GPU_MAX_HW_QUEUES=$q python tools/stream_map_probe.py 0 2>&1 | grep dummy; done
python tools/stream_map_probe.py 0 2>&1 | grep dummy
} | tee gpurun_out/r05t/stream_map.txt
