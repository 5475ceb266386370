# kernel-trace profile of the sequential bench against alternative builds of the library: bash tools/gpu_variants.sh dir
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in $(ls $1/*.so); do
  t=$(basename $f .so)
  SSDR_AL_LIBRARY=$GRAFT_REPO_ROOT/$f rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_$t -o prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > gpurun_out/var_$t.log 2>&1
done
