python -m pytest tests -m gpu -x -q 2>&1 | tail -2; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms'])
"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
