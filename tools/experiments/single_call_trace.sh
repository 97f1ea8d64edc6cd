#!/bin/bash
# HIP API / kernel / copy breakdown of the single-call (host-pointer) searches: where the time of a call goes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-tests/tools/time_search_projection.py}
mkdir -p $R/gpurun_out/sct
python3 $R/$T > $R/gpurun_out/sct/plain.txt 2>&1
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/sct_prof -o sbp -- python3 $R/$T > $R/gpurun_out/sct/traced.txt 2>&1
cat $R/gpurun_out/sct/plain.txt
tail -5 $R/gpurun_out/sct/traced.txt
find /tmp/sct_prof -name '*stats*.csv' | while read f; do echo "== $f"; head -22 $f; cp $f $R/gpurun_out/sct/; done
