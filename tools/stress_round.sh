#!/bin/bash
# Randomised parity sweeps of the final build against the CPU oracle, with a KEPT record: gpurun_out/<tag>/stress_<name>.txt holds the
# commit, the command and the sweep's own summary line; copy them to profiles/<tag>_stress_<name>.txt.  Exits non-zero - loudly - when a
# sweep finds a mismatch or dies (a Python traceback is a failure, not a result).
# Usage on the GPU box: bash tools/stress_round.sh <tag> [scale]      (scale 1 = ~6 minutes of sweeps)
tag=${1:-r04}; scale=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
commit=$(cat .git_commit 2>/dev/null || echo "see the profiles/ commit")
fail=0
run() {  # name, cases, seed, script
  local name=$1 cases=$(( $2 * scale )) seed=$3 script=$4 f=$o/stress_$1.txt
  echo "# $script $cases $seed   (sources sha256 $(sha256sum wut-cuda-orb-slam3_amd/csrc/orbx_kernels.hip wut-cuda-orb-slam3_amd/csrc/orbx_match.hip wut-cuda-orb-slam3_amd/csrc/orbx_api.cpp | sha256sum | cut -c1-16), $(date -u +%Y-%m-%dT%H:%MZ))" > $f
  timeout -k 10 900 python3 $script $cases $seed > $o/stress_$name.log 2>&1; local rc=$?
  grep -E "MISMATCH|Traceback|Error" $o/stress_$name.log | head -20 >> $f
  tail -n 1 $o/stress_$name.log >> $f
  echo "exit $rc" >> $f
  if [ $rc -ne 0 ]; then echo "!!!! STRESS SWEEP $name FAILED (exit $rc) !!!!"; tail -n 15 $o/stress_$name.log; fail=1; fi
  cat $f
}
run parity 150 11 tests/tools/stress_parity.py
STRESS_BATCH=20 run parity_batch 60 21 tests/tools/stress_parity.py   # batch handles: 360-px tiles, margin bands, capped group queue, multi-wave quadtree
run matcher 120 12 tests/tools/stress_matcher.py
run stereo_frames 30 16 tests/tools/stress_stereo_frames.py   # the batch forms behind a stereo rig over 1 .. 129 pairs
run search_init 60 13 tests/tools/stress_search_init.py
run tracking_frames 30 17 tests/tools/stress_tracking_frames.py   # SearchByProjection frame / map-point batch forms over 1 .. 260 frames
run widening 80 14 tests/tools/stress_widening.py
run keyframe 60 15 tests/tools/stress_keyframe.py
if [ $fail -ne 0 ]; then echo "stress_round: AT LEAST ONE SWEEP FAILED"; exit 1; fi
echo "stress_round: all sweeps clean"
