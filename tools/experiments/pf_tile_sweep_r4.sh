#!/bin/bash
# k_pyr_fast: tile width beyond the default (ORBX_TILE_W, with and without the 40 KB LDS fit) and 256 / 384 threads per workgroup
# (liborbx_256.so / liborbx_384.so, built with -DPF_NT=...): pyramid+FAST stage ms and ms per batch of the c2 bench, two runs each.
# Usage on the GPU box: bash tools/experiments/pf_tile_sweep_r4.sh
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python3 bench.py --batches-per-step 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms']['pyramid+FAST'], d['ms_per_step'], end='   ')"; }
sweep() {  # library, list of "tile_w lds_fit"
  local lib=$1; shift
  [ -f wut-cuda-orb-slam3_amd/$lib ] || return
  export ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$lib
  for cfg in "$@"; do
    set -- $cfg
    [ $1 = 0 ] && unset ORBX_TILE_W || export ORBX_TILE_W=$1
    export ORBX_PF_LDS_FIT=$2
    echo -n "$lib tile_w $1 lds_fit $2: "; run; run; echo
  done
}
sweep liborbx.so "0 1" "340 0" "360 0" "380 0" "400 0" "0 1"
sweep liborbx_384.so "0 1" "250 1" "288 0" "320 0"
sweep liborbx_256.so "140 1" "150 1" "160 1" "175 1" "190 1"
