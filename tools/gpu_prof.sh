# kernel-trace profile of the sequential bench -> gpurun_out/$1/ ; prints nothing (read the csv afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1 -o prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > gpurun_out/$1.log 2>&1
