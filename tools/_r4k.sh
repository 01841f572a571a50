mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_production_kernels.py -m gpu -x -q > gpurun_out/r4k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k/pytest.log
tail -6 gpurun_out/r4k/pytest.log
