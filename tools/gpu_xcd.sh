cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/xcd; mkdir -p $OUT
python -m pytest tests/test_randla.py tests/test_pipeline.py tests/test_tile.py tests/test_knn.py -m gpu -x -q 2>&1 | tail -2
bash tools/gpu_netseq.sh bf16x3 2>&1 | grep -E "lfa_|gather_max|grid_search|tile_gather|tail_|sum "
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/pf.err
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open("$OUT/pf/pf_counter_collection.csv")):
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("ssdr::","").replace("(anonymous namespace)::","")
    a=acc[n]; a[0]+=1; a[1]+=float(r["Counter_Value"])*1024*2
for k,v in sorted(acc.items(), key=lambda kv:-kv[1][1])[:16]:
    print("%-50s launches %4d  fetch x2 per launch %8.1f MB" % (k[:50], v[0], v[1]/v[0]/1e6))
PY
rm -rf $OUT/pf
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
