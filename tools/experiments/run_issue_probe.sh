#!/bin/bash
# Generates the issue probes into a temp directory, builds them for gfx950 and runs them (GPU box):
#   bash tools/experiments/run_issue_probe.sh [probe-name-substring] > gpurun_out/issue_probe.txt
# The generated source (52 k lines) is scratch, not tracked; the results of round 4 are profiles/r04_issue_probe_v*.txt.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
exe=$(python3 tools/experiments/gen_issue_probe.py --build | tail -1)
timeout -k 10 300 "$exe" "$@"
