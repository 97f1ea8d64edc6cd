#!/bin/bash
# round 4, GPU job 6: A/B of the s_nop pass (tools/sq_slowmode.py) on the whole extraction step, then the GPU tests on the fixed build
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4job6; mkdir -p $o
for v in nofix fix nofix fix; do
  lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx.so; [ $v = nofix ] && lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx_nofix.so
  ORBX_LIB=$lib timeout -k 10 200 python3 bench.py --batches-per-step 1 --no-cpu-baseline --no-pcie 2>$o/err_$v.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['stages_ms'])"
done
for c in c3 c4 c5; do
for v in nofix fix; do
  lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx.so; [ $v = nofix ] && lib=$GRAFT_REPO_ROOT/wut-cuda-orb-slam3_amd/liborbx_nofix.so
  ORBX_LIB=$lib timeout -k 10 200 python3 bench.py --config $c --batches-per-step 1 --no-cpu-baseline --no-pcie 2>>$o/err_$v.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('matcher',{}).get('ms_per_call'))"
done
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc" >> $o/pytest_gpu.log; tail -5 $o/pytest_gpu.log
