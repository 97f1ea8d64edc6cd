#!/bin/bash
# GaussianBlur and quadtree stage times of a workload (CFG, default c2) together, and each (nearly) alone: ORBX_QT_STOP=1 cuts the
# quadtree short after its gather (timing only - the keypoints are wrong), ORBX_NO_BLUR=1 skips the blur launches.
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/blur
for mode in both blur_alone qt_alone; do
  case $mode in both) env_="";; blur_alone) env_="ORBX_QT_STOP=1";; qt_alone) env_="ORBX_NO_BLUR=1";; esac
  env $env_ timeout -k 10 200 python bench.py --config ${CFG:-c2} --no-cpu-baseline --no-pcie > gpurun_out/blur/${CFG:-c2}_$mode.json 2> gpurun_out/blur/err.txt || { tail -3 gpurun_out/blur/err.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/blur/${CFG:-c2}_$mode.json'))
print('$mode', 'fps', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
done
