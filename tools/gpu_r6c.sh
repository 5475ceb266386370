#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_al_round.py tests/test_select.py tests/test_composition.py tests/test_knn.py -x -q -m gpu > gpurun_out/r6c_tests.txt 2>&1; tail -5 gpurun_out/r6c_tests.txt
timeout 900 python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err; tail -3 gpurun_out/r6c_bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r6c_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms']); a=d['al_round']; print({k:a[k] for k in ('ms','Mpoints_per_s','inference_ms','selection_ms','fps_ms')}); r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us'], r['achieved'], r['unit']); print(json.dumps(r['families'], indent=0)); print({k:(v['ms_per_step'], v.get('frac_of_mfma_peak'), v.get('algorithmic_GFLOP_per_tile')) for k,v in r['others'].items() if k.startswith(('lfa32','dense'))})"
