#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python tools/preset_select.py 0.25 2> $O/s15_preset.err | grep -v amdgpu.ids > $O/s15_preset_select.jsonl; tail -3 $O/s15_preset.err; tail -c 1500 $O/s15_preset_select.jsonl
