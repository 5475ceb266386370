#!/bin/bash
# PMC passes restricted to the network kernels (fast: the other kernels run unprofiled): tools/gpu_pmc_net.sh [regex]
RX=${1:-lfa32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_net
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 170 rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/p$i -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/p$i.err
  cp $OUT/p$i/p_counter_collection.csv $OUT/pass$i.csv 2>/dev/null; rm -rf $OUT/p$i
done
python3 - <<PY
import csv, collections, glob
d = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in sorted(glob.glob("$OUT/pass*.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("ssdr::", "").split("(")[0]
        d[n][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(n, f)].add(r["Dispatch_Id"])
for n in sorted(d):
    k = max(len(v) for (m, f), v in nd.items() if m == n)
    print(n, "dispatches", k)
    print("   ", {c.replace("_sum", ""): "%.4g" % (v / k) for c, v in sorted(d[n].items())})
PY
