#!/bin/bash
# large batches: one wave per quadtree instance (default) against workgroups of 8 waves (ORBX_QT_MW=1), c2 and c4
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python3 bench.py --config $1 --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['stages_ms'])"; }
for c in c2 c4; do
  echo -n "$c default: "; run $c
  echo -n "$c ORBX_QT_MW=1: "; ORBX_QT_MW=1 run $c
  echo -n "$c ORBX_QT_MW=1 ORBX_QT_PER_LEVEL=0: "; ORBX_QT_MW=1 ORBX_QT_PER_LEVEL=0 run $c
  echo -n "$c default: "; run $c
done
