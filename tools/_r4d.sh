mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -x -q > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/pytest.log
tail -4 gpurun_out/r4d/pytest.log
E="python bench.py --emulate-world 8 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs"
$E > gpurun_out/r4d/emu.json 2> gpurun_out/r4d/emu.err
D3P_XCHG_SELF_TRIP=1 $E > gpurun_out/r4d/emu_trip.json 2> gpurun_out/r4d/emu_trip.err
D3P_XCHG_W8=1 $E > gpurun_out/r4d/emu_w8.json 2> gpurun_out/r4d/emu_w8.err
$E --sampler poisson > gpurun_out/r4d/emu_poisson.json 2> gpurun_out/r4d/emu_poisson.err
D3P_DBG=32 $E > gpurun_out/r4d/emu_anat.json 2> gpurun_out/r4d/emu_anat.err
D3P_XCHG_SELF_TRIP=1 D3P_DBG=32 $E > gpurun_out/r4d/emu_trip_anat.json 2> gpurun_out/r4d/emu_trip_anat.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r4d/bench_driver.json 2> gpurun_out/r4d/bench_driver.err
grep -o "kernel_us_per_step\": [0-9.]*" gpurun_out/r4d/*.json
grep -o "\"steps_per_sec\": [0-9.]*" gpurun_out/r4d/*.json | head -20
