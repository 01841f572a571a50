set -ex
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/${PROFILE_TAG:-r01e}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r01 -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-large-batch > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-large-batch > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-large-batch > /dev/null 2> $O/pmc_write.err
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json 330 > /dev/null 2> $O/pmc_traffic.err
cp $O/pmc_traffic.json $R/profiles/traffic_latest.json 2>/dev/null
python3 bench.py --steps 3000 --warmup 300 > $O/bench.json 2> $O/bench.err
ls -R $O | head -40
