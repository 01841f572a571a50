mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_production_kernels.py tests/test_dist.py -m gpu -x -q > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c/pytest.log
tail -4 gpurun_out/r4c/pytest.log
E="python bench.py --emulate-world 8 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs"
for v in 0 4096; do D3P_DBG=$v $E > gpurun_out/r4c/emu_dbg$v.json 2> gpurun_out/r4c/emu_dbg$v.err; done
for v in 0 4096; do D3P_XCHG_SELF_TRIP=1 D3P_DBG=$v $E > gpurun_out/r4c/emu_trip_dbg$v.json 2> gpurun_out/r4c/emu_trip_dbg$v.err; done
D3P_DBG=32 $E > gpurun_out/r4c/emu_anat.json 2> gpurun_out/r4c/emu_anat.err
D3P_XCHG_SELF_TRIP=1 D3P_DBG=32 $E > gpurun_out/r4c/emu_trip_anat.json 2> gpurun_out/r4c/emu_trip_anat.err
D3P_DBG=$((32+0x700)) $E > gpurun_out/r4c/emu_waves.json 2> gpurun_out/r4c/emu_waves.err
grep -o "kernel_us_per_step\": [0-9.]*" gpurun_out/r4c/emu*.json
grep -v amdgpu.ids gpurun_out/r4c/emu_anat.err | head -24
grep -v amdgpu.ids gpurun_out/r4c/emu_trip_anat.err | head -24
grep -v amdgpu.ids gpurun_out/r4c/emu_waves.err | head -8
