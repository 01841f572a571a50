# Round profile of the headline bench (run on the GPU box through gpurun): rocprofv3 kernel stats, then the two PMC
# passes for HBM traffic (separate passes, MI355X_MICROARCH.md section HBM), then the unprofiled bench line.
# usage: PROFILE_TAG=r02 bash tools/profile_round.sh    (outputs under gpurun_out/$PROFILE_TAG; copy what is judged to profiles/)
set -ex
R=$(cd "$(dirname "$0")/.." && pwd)
T=${PROFILE_TAG:-r02}
O=$R/gpurun_out/$T
COMMIT=${PROFILE_COMMIT:-unknown}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $T -- python3 $R/bench.py --steps 2048 --warmup 256 --no-cpu-baseline --no-large-batch --no-extra-legs > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 384 --warmup 128 --no-cpu-baseline --no-large-batch --no-extra-legs > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --steps 384 --warmup 128 --no-cpu-baseline --no-large-batch --no-extra-legs > /dev/null 2> $O/pmc_write.err
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json 512 ${T}_pmc_traffic.json $COMMIT > /dev/null 2> $O/pmc_traffic.err
python3 bench.py > $O/bench.json 2> $O/bench.err
ls -R $O | head -40
