#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu -k "dwconv or depthwise or skblock" 2>&1 | tail -3
for args in "640 24 7 f16x2 f16out" "324 24 15 f16x2 f16out single" "256 24 15 f16x2 f16out single" "384 8 15 f16x2 f16out" "640 3 7 f16x2 f16out" "324 3 15 f16x2 f16out single"; do
  python tools/dwconv_one.py $args 2>&1 | tail -1
done
