# developer A/B: production-kernel oracle tests, then in-kernel time per step of the chained launch in a few variants
set -e
python -m pytest tests/test_gpu_production_kernels.py -x -q -m gpu > gpurun_out/prod_tests.log 2>&1 || { tail -30 gpurun_out/prod_tests.log; exit 1; }
tail -3 gpurun_out/prod_tests.log
rm -f gpurun_out/ab.jsonl
for i in 1 2; do
TAG=w16 python tools/time_chained.py 2048 >> gpurun_out/ab.jsonl
TAG=w8 D3P_CHAIN_W8=1 python tools/time_chained.py 2048 >> gpurun_out/ab.jsonl
done
TAG=w16_icpt python tools/time_chained.py 2048 4096 512 1 >> gpurun_out/ab.jsonl
TAG=w16_32k python tools/time_chained.py 512 32768 >> gpurun_out/ab.jsonl
TAG=w16_8k python tools/time_chained.py 1024 8192 >> gpurun_out/ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/ab.jsonl'):
    d=json.loads(l); print(d['tag'], d['kernel_us_per_step'], d['wall_us_per_step'][1], d['final_loss'])
PY
D3P_ANATOMY_ROWS=1 D3P_DBG=32 python tools/time_chained.py 512 > gpurun_out/an16.json 2> gpurun_out/an16.txt
grep -v amdgpu.ids gpurun_out/an16.txt
