mkdir -p gpurun_out/r4n
python -m pytest tests -m gpu -x -q > gpurun_out/r4n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4n/pytest.log
tail -4 gpurun_out/r4n/pytest.log
