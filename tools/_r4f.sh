mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_minibatch.py tests/test_dist.py -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4f/pytest.log
tail -30 gpurun_out/r4f/pytest.log
