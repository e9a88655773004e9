#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O
timeout 3000 python tests/analysis/preset_select_v2.py > $O/r05_preset_select.jsonl 2> $O/sel.err; tail -3 $O/sel.err; tail -c 2500 $O/r05_preset_select.jsonl
