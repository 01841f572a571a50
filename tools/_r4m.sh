mkdir -p gpurun_out/r4m
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux-workloads --no-large-batch > gpurun_out/r4m/b$i.json 2> gpurun_out/r4m/b$i.err; python -c "
import json; d=json.load(open('gpurun_out/r4m/b$i.json')); print(d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_us_per_step'], d['steady_state']['steps_per_sec'])"; done
