#!/bin/bash
# The rest of the evidence set under profiles/ (after tools/profile_round.sh): phase counters and SQ counters of k_pyr_fast, single-frame
# and stereo-frame latency, kernel timeline of one call, matcher / tracking / initialization batch timings.
# Usage on the GPU box: bash tools/profile_extras.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
bash tools/pmc_phases.sh > $o/pyr_fast_phase_counters.txt 2>&1
bash tools/pmc.sh > $o/sq_counters.txt 2>&1
timeout -k 10 120 python tools/time_single_frame.py 2>&1 | grep -v amdgpu.ids > $o/single_frame.txt
bash tools/experiments/single_frame_trace.sh > $o/single_frame_trace.txt 2>&1
timeout -k 10 200 python tools/time_stereo_pair.py 2>&1 | grep -v amdgpu.ids > $o/stereo_pair.txt
timeout -k 10 200 python tools/time_stereo_pair.py 752 480 1200 2>&1 | grep -v amdgpu.ids >> $o/stereo_pair.txt
timeout -k 10 300 python tools/time_matcher.py > $o/matcher_knn2.json 2> $o/matcher_knn2.err
timeout -k 10 300 python tests/tools/time_tracking_batch.py > $o/tracking_batch.json 2> $o/tracking_batch.err
timeout -k 10 300 python tests/tools/time_search_init_batch.py > $o/search_init_batch.json 2> $o/search_init_batch.err
tail -n 3 $o/single_frame.txt $o/stereo_pair.txt; tail -5 $o/single_frame_trace.txt
