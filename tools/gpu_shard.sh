cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_distributed.py -m gpu -x -q 2>&1 | tail -3
STEPS=60 bash tools/gpu_emulate_world.sh
SSDR_BENCH_FORCE_DIST=1 SSDR_EMULATE_WORLD=8 python3 tools/hosttime.py 2>&1 | tail -9
