# per-stage line of the sequential bench, optionally under an env override:  bash tools/gpu_quick2.sh [VAR=1 ...]
for v in "$@"; do export "$v"; done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms'])"
