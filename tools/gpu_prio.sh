#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in 0 1 -1 1 0; do
  if [ "$P" = "0" ]; then unset SSDR_SELECT_PRIORITY; else export SSDR_SELECT_PRIORITY=$P; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('selection priority $P:', d['value'], d['ms_per_step'])"
done
