mkdir -p gpurun_out/r4j
python bench.py --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4j/emu20.json 2> gpurun_out/r4j/emu20.err
python bench.py --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r4j/emu20_cold.json 2> gpurun_out/r4j/emu20_cold.err
grep -o "\"steps_per_sec\": [0-9.]*\|kernel_us_per_step\": [0-9.]*\|\"ms_per_step\": [0-9.]*" gpurun_out/r4j/*.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/emu20 -o emu20 -- python3 /root/repo/bench.py --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; cd /root/repo; cp $(find /tmp/emu20 -name "*kernel_stats.csv" | head -1) gpurun_out/r4j/emu20_kernel_stats.csv; cut -c1-150 gpurun_out/r4j/emu20_kernel_stats.csv | head -12
