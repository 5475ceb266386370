#!/bin/bash
# PMC passes over the network kernels of one sequential step (wave cycles, waits, instruction mix, texture-addresser stalls)
bash tools/gpu_pmc.sh bf16x3 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES" \
   "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA" \
   "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > /dev/null 2>&1
D=$GRAFT_REPO_ROOT/gpurun_out/pmc_bf16x3
python3 tools/pmc_table.py $D lfa dense tail gather_max
python3 - <<PY
import csv, collections
for f in ("pass2", "pass3"):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open("$D/%s.csv" % f)):
        n = r["Kernel_Name"].replace("void ", "").replace("ssdr::", "").split("(")[0]
        if "lfa" in n or "dense" in n: d[n][r["Counter_Name"]] += float(r["Counter_Value"])
    for n in sorted(d): print(f, n[:40], {k: "%.3g" % v for k, v in d[n].items()})
PY
