mkdir -p gpurun_out/r4g
E="python bench.py --emulate-world 8 --rows-per-gpu 1250000 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs --sampler poisson"
$E > gpurun_out/r4g/emu_poisson_sharded.json 2> gpurun_out/r4g/emu_poisson_sharded.err
D3P_POISSON_FULL_MASK=1 $E > gpurun_out/r4g/emu_poisson_fullmask.json 2> gpurun_out/r4g/emu_poisson_fullmask.err
python bench.py --rows-per-gpu 10000000 --steps 2048 --warmup 256 --no-cpu-baseline --no-extra-legs --sampler poisson > gpurun_out/r4g/single_poisson_1e7.json 2> gpurun_out/r4g/single_poisson_1e7.err
grep -o "\"steps_per_sec\": [0-9.]*\|kernel_us_per_step\": [0-9.]*\|\"ms_per_step\": [0-9.]*" gpurun_out/r4g/*.json
tail -2 gpurun_out/r4g/*.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ps -o ps -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --rows-per-gpu 1250000 --steps 512 --warmup 128 --no-cpu-baseline --no-extra-legs --sampler poisson > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; find /tmp/prof_ps -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4g/emu_poisson_sharded_kernel_stats.csv; head -12 gpurun_out/r4g/emu_poisson_sharded_kernel_stats.csv | cut -c1-200
