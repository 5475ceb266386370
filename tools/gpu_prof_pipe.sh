# kernel-trace profile of the PIPELINED bench -> gpurun_out/$1/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$1 -o prof -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/$1.log 2>&1
