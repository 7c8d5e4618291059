#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  \|^$" | tail -6
true
