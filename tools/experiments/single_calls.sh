#!/bin/bash
# latencies of the single-call (host-pointer) entry points: the timing tools of the matcher side in one run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/single_calls
O=gpurun_out/single_calls/times.txt
: > $O
python3 tests/tools/time_search_projection.py >> $O 2>&1
python3 tests/tools/time_keyframe_searches.py >> $O 2>&1
python3 tools/time_search_init.py >> $O 2>&1
python3 tools/time_stereo_pair.py >> $O 2>&1
python3 tools/time_matcher.py >> $O 2>&1
cat $O
