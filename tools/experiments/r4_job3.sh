#!/bin/bash
# round 4, GPU job 3: mixing probes + per-phase counters of the new k_pyr_fast
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4job3; mkdir -p $o
bash tools/experiments/run_issue_probe.sh > $o/issue_probe.txt 2>&1; echo "issue_probe exit=$?"
bash tools/pmc_phases.sh > $o/pyr_fast_phase_counters.txt 2>&1
cat $o/pyr_fast_phase_counters.txt
