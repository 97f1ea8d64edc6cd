export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "host_batch or sharded or small or single or graph or lapping" > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3b/pytest.log
timeout -k 10 300 python tools/time_host_batch.py > gpurun_out/r3b/host_batch.txt 2>&1; cat gpurun_out/r3b/host_batch.txt
for s in 4 8; do for fd in 0 1; do echo "sub $s feeder $fd"; ORBX_HOST_SUBBATCHES=$s ORBX_HOST_FEEDER=$fd timeout -k 10 200 python tools/time_host_batch.py 2>&1 | grep orbx_extract_batch; done; done
