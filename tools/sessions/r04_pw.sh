cd "$GRAFT_REPO_ROOT"; timeout 300 python tools/pw_b_format.py 2>&1 | tail -6
