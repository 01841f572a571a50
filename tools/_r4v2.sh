set -e
O=$PWD/gpurun_out/r4v; mkdir -p $O
R=$PWD
python tools/time_vae_step.py
python tools/time_vae_step.py
python tools/time_vae_step.py
python tools/time_vae_step.py
python tools/time_vae_step.py 200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_stats -o vae -- python3 $R/tools/time_vae_step.py > $O/vae_stats.log 2>&1
cd $R
python tools/kernel_timeline.py $O/vae_stats k_vae_keys
