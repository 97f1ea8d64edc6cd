cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 500 python -m pytest tests/test_matcher_gpu.py tests/test_search_projection_gpu.py tests/test_search_init_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3a/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; cat gpurun_out/r3a/bench.json
timeout -k 10 300 python bench.py --gpus 2 --steps 5 --no-cpu-baseline > gpurun_out/r3a/bench2.json 2> gpurun_out/r3a/bench2.err; echo "bench2 rc=$?"; cat gpurun_out/r3a/bench2.json; tail -3 gpurun_out/r3a/bench2.err
timeout -k 10 300 python tools/time_host_batch.py > gpurun_out/r3a/host_batch.txt 2>&1; cat gpurun_out/r3a/host_batch.txt
