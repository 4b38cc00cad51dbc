set -e
O=gpurun_out/r2/final
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
python bench.py --workload config4 --no-orbit > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 300 $O/bench_config4.json
python tests/tools/config_bench.py > $O/all_configs.txt 2>&1; tail -3 $O/all_configs.txt
MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/timeline.py > $O/timeline.txt 2>&1
MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/phase_profile.py 20spheres 1920 1080 > $O/phase_shares.txt 2>&1
MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/phase_profile.py 20spheres 3840 2160 >> $O/phase_shares.txt 2>&1
MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/phase_profile.py 20spheres 7680 4320 >> $O/phase_shares.txt 2>&1
python tools/flythrough_bench.py > $O/flythrough.txt 2>&1; tail -3 $O/flythrough.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-orbit > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/kt.err)
bash tools/pmc_profile.sh $O/pmc
python tools/pmc_summarize.py $O/pmc > $O/pmc_summary.txt 2>&1; tail -5 $O/pmc_summary.txt
find $O/pmc $O/kt -name "*.csv" -size +3M -delete
