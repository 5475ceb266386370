#!/bin/bash
# counters of the kernels whose name contains $1 in one sequential bench step: tools/gpu_pmc_kernel.sh chamfer [extra env]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -o b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/b.err
python3 - "$1" $OUT/a/a_counter_collection.csv $OUT/b/b_counter_collection.csv <<'P'
import csv, sys, collections
pat = sys.argv[1]
for f in sys.argv[2:]:
    acc = collections.defaultdict(float); n = collections.Counter()
    try:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    except FileNotFoundError:
        print("missing", f); continue
    for k in acc: print(f"{k:32s} {acc[k]:16.0f}  ({n[k]} rows)")
P
