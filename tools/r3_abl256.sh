#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python tools/bench_wino2d.py --variants wino42 >/dev/null 2>&1  # warm the box
for rep in 1 2; do ABLS="256 64" bash tools/ablate_wino42.sh; done
