#!/bin/bash
# the sharded code path (RCCL, world size 1) carrying the FPS load of N ranks (SSDR_EMULATE_WORLD): step time, and whether the FPS kernels of consecutive batches overlap
OUT=$GRAFT_REPO_ROOT/gpurun_out/emuworld
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1
for N in 1 2 4 8; do
  SSDR_EMULATE_WORLD=$N python3 bench.py --steps 100 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPS load of $N rank(s):', d['value'], 'Mpoints/s per GPU,', d['ms_per_step'], 'ms per step')"
done
SSDR_EMULATE_WORLD=4 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/kt.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
f=sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if "fps_coop" in r["Kernel_Name"] or "fps_block" in r["Kernel_Name"])
f=f[6:-3]
ov=sum(max(0,min(f[i][1],f[i+1][1])-f[i+1][0]) for i in range(len(f)-1)); tot=sum(e-s for s,e,_ in f[:-1])
print("N = 4: FPS kernels %d, mean %.2f ms, overlap with the next one %.0f%%, start to start %.2f ms, queues %s" % (len(f), tot/max(1,len(f)-1)/1e6, 100*ov/max(1,tot), (f[-1][0]-f[0][0])/max(1,len(f)-1)/1e6, [x[2] for x in f[:6]]))
PY
rm -rf $OUT/kt
