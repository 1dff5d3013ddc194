cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "winograd42" 2>&1 | tail -8
for nw in 2 1; do echo "=== NW=$nw B=128"; GDDIM_WINO42_NW=$nw python tools/bench_wino2d.py --batch 128 --variants wino42 --iters 10 2>&1 | sed -e 's/wino42: vs direct .* |//' | tail -13; done
for nw in 2 1; do echo "=== NW=$nw B=256"; GDDIM_WINO42_NW=$nw python tools/bench_wino2d.py --batch 256 --variants wino42 --iters 10 2>&1 | sed -e 's/wino42: vs direct .* |//' | tail -13; done
