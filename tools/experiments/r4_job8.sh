#!/bin/bash
# round 4, GPU job 8: per-phase counters of k_pyr_fast with the s_nop pass (compare with profiles/r04a_*, the same sources without it) + stage-time A/B
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4job8; mkdir -p $o
for v in nofix fix nofix fix; do
  lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx.so; [ $v = nofix ] && lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx_nofix.so
  ORBX_LIB=$lib timeout -k 10 200 python3 bench.py --batches-per-step 1 --no-cpu-baseline --no-pcie 2>$o/err_$v.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stages_ms'])"
done
bash tools/pmc_phases.sh > $o/pyr_fast_phase_counters_fix.txt 2>&1
cat $o/pyr_fast_phase_counters_fix.txt
