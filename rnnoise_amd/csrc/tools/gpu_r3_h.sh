#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; export TMPDIR=/tmp
python tools/nn_one_taps.py 1 2>&1 | grep -v amdgpu
true
