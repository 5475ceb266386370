#!/bin/bash
# The sharded code path (RCCL at world size 1: device-side rule, all-gather, replicated global FPS) carrying the FPS load of N ranks on ONE GPU
# (SSDR_EMULATE_WORLD=N: the candidate rows are taken N times in front of the FPS, N x the picks) — a projection of the replicated chain's cost,
# with none of the collectives' latency in it.  Writes gpurun_out/emuworld/emulated_world.json (copied to profiles/ by hand) and prints the host time
# the step spends inside the selection's enqueue.
OUT=$GRAFT_REPO_ROOT/gpurun_out/emuworld
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1
rm -f $OUT/lines.jsonl
for N in 1 2 4 8; do
  SSDR_EMULATE_WORLD=$N python3 bench.py --steps ${STEPS:-100} --no-cpu-baseline --no-al-round 2>/dev/null | grep '^{"metric"' >> $OUT/lines.jsonl
done
python3 - <<PY
import json
L=[json.loads(l) for l in open("$OUT/lines.jsonl")]
base=L[0]["value"]
res={"command": "MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1 SSDR_EMULATE_WORLD=N python3 bench.py --steps ${STEPS:-100} --no-cpu-baseline (tools/gpu_emulate_world.sh)",
     "what": "one MI355X running the sharded code path at world size 1 with the replicated global FPS carrying N x the candidate rows and N x the picks; no collective latency of N ranks is in these numbers: a projection, not a scaling curve",
     "runs": [{"emulated_ranks": d["config"].get("emulated_world", 1), "Mpoints_per_s_per_gpu": d["value"], "ms_per_step": d["ms_per_step"], "selection_rule": d["config"].get("selection_rule"),
               "whole_job_efficiency": round(d["value"] / base, 3)} for d in L]}
json.dump(res, open("$OUT/emulated_world.json", "w"), indent=1)
for r in res["runs"]: print(r)
PY
