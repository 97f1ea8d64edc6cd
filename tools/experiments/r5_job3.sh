#!/bin/bash
# round 5, GPU job 3: what does the resize cost INSIDE the full kernel (phase ablation), old byte-read resize vs the dword-window resize,
# and the SQ counters of both builds
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5job3; mkdir -p $o
for lib in liborbx_nowin.so liborbx.so; do
  echo "== $lib" | tee -a $o/ablate.txt
  for ph in 15 14 13 11 7; do
    echo -n "phases=$ph " | tee -a $o/ablate.txt
    ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$lib ORBX_PF_PHASES=$ph timeout -k 10 200 python3 bench.py --batches-per-step 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stages_ms'])" | tee -a $o/ablate.txt
  done
done
for lib in liborbx_nowin.so liborbx.so; do
  echo "== $lib"
  ORBX_LIB=$PWD/wut-cuda-orb-slam3_amd/$lib bash tools/pmc.sh 2>&1 | grep "k_pyr_fast" | tee -a $o/pmc_$lib.txt
done
