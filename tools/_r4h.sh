mkdir -p gpurun_out/r4h
python -m pytest tests/test_dist.py -m gpu -x -q -k "simulated_peers or virtual_ranks or poisson" > gpurun_out/r4h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4h/pytest.log
tail -25 gpurun_out/r4h/pytest.log
