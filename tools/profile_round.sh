# Round profile of the headline bench (run on the GPU box through gpurun): rocprofv3 kernel stats, then the two PMC
# passes for HBM traffic (separate passes, MI355X_MICROARCH.md section HBM), then the unprofiled bench line.
# usage: PROFILE_TAG=r03 bash tools/profile_round.sh    (outputs under gpurun_out/$PROFILE_TAG; copy what is judged to profiles/)
set -ex
R=$(cd "$(dirname "$0")/.." && pwd)
T=${PROFILE_TAG:-r06}
O=$R/gpurun_out/$T
COMMIT=${PROFILE_COMMIT:-unknown}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $T -- python3 $R/bench.py --steps 2048 --warmup 256 --no-cpu-baseline --no-large-batch --no-extra-legs > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 384 --warmup 128 --no-cpu-baseline --no-large-batch --no-extra-legs > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --steps 384 --warmup 128 --no-cpu-baseline --no-large-batch --no-extra-legs > /dev/null 2> $O/pmc_write.err
# every leg of the driver's command (headline, steady state, N = 1e7, Poisson sampler, batch 32768, mixture model, both VAE shapes)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_all -o ${T}_all -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_all_legs_under_rocprof.json 2> $O/stats_all.err
# the step kernel's VALU instruction count (bench.py roofline.valu): its own pass
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_valu -o valu -- python3 $R/bench.py --steps 384 --warmup 128 --no-cpu-baseline --no-large-batch --no-extra-legs > /dev/null 2> $O/pmc_valu.err
# FETCH_SIZE / WRITE_SIZE per access width on a known byte count (1 GiB copies at 16, 8, 4 bytes per lane)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o fetch -- python3 $R/tools/probes/fetch_calibration.py > /dev/null 2> $O/cal_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o write -- python3 $R/tools/probes/fetch_calibration.py > /dev/null 2> $O/cal_write.err
cd $R
python3 tools/probes/fetch_calibration_report.py $O/cal_fetch $O/cal_write $O/fetch_calibration.json > /dev/null 2> $O/fetch_calibration.err || true
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json 512 ${T}_pmc_traffic.json $COMMIT 4096 512 $O/fetch_calibration.json > /dev/null 2> $O/pmc_traffic.err
python3 tools/valu_pmc.py $O/pmc_valu $O/logreg_valu_pmc.json 128 $COMMIT > /dev/null 2> $O/valu_pmc.err
# (bench.py reads the newest profiles/rNN_{pmc_traffic,logreg_valu_pmc}.json: this run's, for the unprofiled line below)
cp $O/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json
cp $O/logreg_valu_pmc.json $R/profiles/${T}_logreg_valu_pmc.json
python3 bench.py > $O/bench.json 2> $O/bench.err
# what is judged: copies under profiles/ (tracked)
P=$R/gpurun_out/${T}_profiles; rm -rf $P; mkdir -p $P
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $P/${T}_kernel_stats.csv
cp $(find $O/stats_all -name "*kernel_stats.csv" | head -1) $P/${T}_kernel_stats_all_legs.csv
cp $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $P/${T}_pmc_FETCH_SIZE_raw.csv || true
cp $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $P/${T}_pmc_WRITE_SIZE_raw.csv || true
cp $O/pmc_traffic.json $P/${T}_pmc_traffic.json
cp $O/logreg_valu_pmc.json $P/${T}_logreg_valu_pmc.json
cp $O/fetch_calibration.json $P/${T}_fetch_calibration.json || true
cp $(find $O/pmc_valu -name "*counter_collection.csv" | head -1) $P/${T}_pmc_VALU_raw.csv || true
cp $O/bench_under_rocprof.json $P/${T}_bench_under_rocprof.json
cp $O/bench_all_legs_under_rocprof.json $P/${T}_bench_all_legs_under_rocprof.json
cp $O/bench.json $P/${T}_bench.json
ls -la $P
# per-kernel VALU instruction counts of the mixture-model leg (bench.py's gmm roofline uses them) and the VAE step's kernel times
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/gmm_pmc -o gmm -- python3 $R/tools/time_gmm_step.py > $O/gmm_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_stats -o vae -- python3 $R/tools/time_vae_step.py > $O/vae_stats.log 2>&1
cd $R
python3 tools/pmc_kernel_means.py $P/${T}_gmm_pmc.json $O/gmm_pmc > /dev/null
cp $(find $O/vae_stats -name "*kernel_stats.csv" | head -1) $P/${T}_vae_kernel_stats.csv
ls -la $P
