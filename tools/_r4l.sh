mkdir -p gpurun_out/r4l
python -m pytest tests/test_dist.py tests/test_gpu_production_kernels.py tests/test_gpu_configs.py tests/test_gpu_dpsvi.py -m gpu -x -q > gpurun_out/r4l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4l/pytest.log
tail -5 gpurun_out/r4l/pytest.log
python tools/time_short_runs.py 20 30 2>&1 | grep -v amdgpu.ids | grep "timing off"
python bench.py --emulate-world 8 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs > gpurun_out/r4l/emu.json 2> gpurun_out/r4l/emu.err
grep -o "\"steps_per_sec\": [0-9.]*\|kernel_us_per_step\": [0-9.]*" gpurun_out/r4l/emu.json | head -3
