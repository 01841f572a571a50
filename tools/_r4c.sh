mkdir -p gpurun_out/r4c
E="python bench.py --emulate-world 8 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs"
for v in 0 0; do D3P_DBG=$v $E > gpurun_out/r4c/emu_dbg$v.json 2> gpurun_out/r4c/emu_dbg$v.err;  grep -o "kernel_us_per_step\": [0-9.]*" gpurun_out/r4c/emu_dbg$v.json; done
for v in 0; do D3P_XCHG_SELF_TRIP=1 D3P_DBG=$v $E > gpurun_out/r4c/emu_trip_dbg$v.json 2> gpurun_out/r4c/emu_trip_dbg$v.err; done
D3P_DBG=32 $E > gpurun_out/r4c/emu_anat.json 2> gpurun_out/r4c/emu_anat.err
grep -o "kernel_us_per_step\": [0-9.]*" gpurun_out/r4c/emu*.json
grep -v amdgpu.ids gpurun_out/r4c/emu_anat.err | head -16
