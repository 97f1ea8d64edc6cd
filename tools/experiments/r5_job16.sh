#!/bin/bash
# round 5, GPU job 16: LDS layout of the other workloads on the margin-band build; tile width against occupancy on c3 / c5 (ORBX_TILE_W)
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job16; mkdir -p $o
for c in "752 480 1200 1024" "512 512 1500 1024" "1920 1080 4000 256"; do echo "== $c"; ORBX_DEBUG=1 python3 tools/experiments/print_geometry.py $c 2>&1 | grep "LDS layout" | sed 's/\[orbx\] //'; done > $o/geom.txt 2>&1
cat $o/geom.txt | cut -c1-60,150-260
run() { env $2 python3 bench.py --config $1 --batches-per-step 1 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['stages_ms']['pyramid+FAST'], d['ms_per_step'], d['value'])"; }
for i in 1 2; do
for c in c3 c5 c4 c2; do
run $c ORBX_TILE_W=360
run $c ORBX_TILE_W=320
run $c ORBX_TILE_W=300
done; done 2>&1 | tee $o/ab.txt
