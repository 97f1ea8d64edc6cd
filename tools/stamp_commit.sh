#!/bin/bash
# Writes the commit the working tree is at into .git_commit (git-ignored, travels to the GPU box: .git does not), so that the records made
# there - profiles/pyr_fast_counters_latest.json (bench.py's traffic_source.taken_at_commit), the stress records - name their build.
# Run in the container before a gpurun call that collects evidence:  bash tools/stamp_commit.sh && gpurun ...
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short=12 HEAD)
if ! git diff --quiet HEAD -- . ':!profiles' ':!gpurun_out'; then c="$c-dirty"; fi
echo "$c" > .git_commit
echo "stamped $c"
