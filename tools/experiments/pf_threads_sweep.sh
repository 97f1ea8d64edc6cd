#!/bin/bash
# k_pyr_fast: threads per workgroup x tile width x kernel organisation (1 phases / 2 a wave per cell).  Prints the pyramid+FAST stage time
# of the c2 bench (ms per 1024 frames).  Usage on the GPU box: bash tools/experiments/pf_threads_sweep.sh
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --batches-per-step 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stages_ms']['pyramid+FAST'], d['ms_per_step'])"; }
for lib in liborbx.so liborbx_256.so; do
  for k in 1 2; do
    for tw in 0 105 140 175 210 245; do
      [ $tw = 0 ] && unset ORBX_TILE_W || export ORBX_TILE_W=$tw
      echo -n "$lib kernel $k tile_w $tw: "
      ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$lib ORBX_PF_KERNEL=$k run
    done
  done
done
