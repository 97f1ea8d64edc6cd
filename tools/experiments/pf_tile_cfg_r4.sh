#!/bin/bash
# default tiles against ORBX_TILE_W=360 (with / without the 40 KB LDS fit) on every bench workload: pyramid+FAST stage ms, ms per batch
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python3 bench.py --config $1 --batches-per-step 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms']['pyramid+FAST'], d['ms_per_step'], end='   ')"; }
for c in c2 c3 c4 c5; do
  for cfg in "0 1" "360 0" "360 1" "330 0" "0 1"; do
    set -- $cfg
    [ $1 = 0 ] && unset ORBX_TILE_W || export ORBX_TILE_W=$1
    export ORBX_PF_LDS_FIT=$2
    echo -n "$c tile_w $1 lds_fit $2: "; run $c; run $c; echo
  done
done
