#!/bin/bash
# Collects the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# then copy gpurun_out/profiles_$1/* into profiles/; tools/pmc_summary.py builds the PMC summary bench.py reads, tools/kernel_table.py the per-kernel
# table (time, HBM rate, vector-issue time, matrix instructions) profiles/$1_kernel_table.md.
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. the bench line itself (default flags = what the driver runs)
python3 bench.py 2> $OUT/${R}_bench_line.err | grep '^{"metric"' > $OUT/${R}_bench_line.json
# 2. kernel trace + stats of the same command (batches in flight) and of the strictly sequential variant
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --no-al-round 2> $OUT/kt.err | grep '^{"metric"' > $OUT/${R}_bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kts -- python3 bench.py --no-cpu-baseline --no-pipeline --no-al-round 2> $OUT/kts.err | grep '^{"metric"' > $OUT/${R}_bench_seq_under_rocprof.json
cp $OUT/kt/kt_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
cp $OUT/kts/kts_kernel_stats.csv $OUT/${R}_bench_seq_kernel_stats.csv
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass), sequential, one step
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-al-round > /dev/null 2> $OUT/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -o pw -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-al-round > /dev/null 2> $OUT/pw.err
cp $OUT/pf/pf_counter_collection.csv $OUT/${R}_pmc_fetch_size.csv
cp $OUT/pw/pw_counter_collection.csv $OUT/${R}_pmc_write_size.csv
# 4. matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) next to GRBM_GUI_ACTIVE (the kernel's cycles, summed over the 8 XCDs): own pass, no tracing besides the counters
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pm -o pm -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --no-al-round > /dev/null 2> $OUT/pm.err
cp $OUT/pm/pm_counter_collection.csv $OUT/${R}_pmc_mfma_busy.csv
# 5. the N > 1 code path (RCCL exchanges on device buffers) on this one GPU, with its kernel stats
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-al-round 2> $OUT/rccl.err | grep '^{"metric"' > $OUT/${R}_bench_rccl_world1_line.json
rm -rf $OUT/kt $OUT/kts $OUT/pf $OUT/pw $OUT/pm
ls -la $OUT
