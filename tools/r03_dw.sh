#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "640 24 7 f16x2 f16out" "640 24 7 f16x2 f16out single"; do
  python tools/dwconv_one.py $args 2>&1 | tail -1
done
