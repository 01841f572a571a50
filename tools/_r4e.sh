mkdir -p gpurun_out/r4e
D3P_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-seconds 2 > gpurun_out/r4e/share2.json 2> gpurun_out/r4e/share2.err; echo "rc=$?"
tail -c 3000 gpurun_out/r4e/share2.json; tail -5 gpurun_out/r4e/share2.err
