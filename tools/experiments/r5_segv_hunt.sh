#!/bin/bash
# host-side crash hunt: test_two_threads_recapturing_all_the_time (+ its neighbours) under rocgdb until a SIGSEGV shows, at most N runs
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
for i in $(seq 1 ${2:-8}); do
  timeout -k 10 200 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex "handle SIG33 nostop noprint" -ex run -ex bt -ex "info threads" --args python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "two_handles_concurrently or two_threads_recapturing or sharded_host_batch" > $o/gdb_$i.txt 2>&1
  if grep -q "SIGSEGV\|SIGABRT\|SIGBUS" $o/gdb_$i.txt; then echo "run $i: CRASH"; grep -v "^\[Thread\|^\[New Thread" $o/gdb_$i.txt | tail -60; exit 0; fi
  echo "run $i: $(grep -o '[0-9]* passed.*' $o/gdb_$i.txt | tail -1)"
done
