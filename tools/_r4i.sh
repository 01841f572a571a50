mkdir -p gpurun_out/r4i
python -m pytest tests/test_gpu_vae.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4i/pytest.log
tail -6 gpurun_out/r4i/pytest.log
python tools/time_vae_step.py > gpurun_out/r4i/vae_time.txt 2>&1; tail -4 gpurun_out/r4i/vae_time.txt
D3P_VAE_NO_EXACT16=1 python tools/time_vae_step.py > gpurun_out/r4i/vae_time_noexact.txt 2>&1; tail -4 gpurun_out/r4i/vae_time_noexact.txt
