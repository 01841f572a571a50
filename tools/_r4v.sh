mkdir -p gpurun_out/r4v
python -m pytest tests/test_gpu_vae.py -m gpu -x -q > gpurun_out/r4v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4v/pytest.log
tail -5 gpurun_out/r4v/pytest.log
for i in 1 2; do
python tools/time_vae_step.py
D3P_VAE_NO_GROUP=1 python tools/time_vae_step.py
done
python tools/time_vae_step.py 200
D3P_VAE_NO_GROUP=1 python tools/time_vae_step.py 200
