#!/bin/bash
# The evidence of a round, on the GPU box: tests, bench, rocprofv3 kernel stats + PMC passes (config 2 and config 4), timeline / phase
# shares of the stamped diagnostic build, orbit, all configs, A/B of the two sphere instantiations, fuzz.  Everything lands under
# gpurun_out/$1 (default r3/final6); the summaries worth judging are copied into profiles/ by hand afterwards.
# A step that is killed at its time limit stops the pass (nothing further touches the GPU after a hang).
# PART=1 tests / bench / A-B / timelines, PART=2 rocprofv3 kernel stats + PMC passes, PART=3 fuzz: one gpurun call each.
# PART=1 needs two libraries under tools/bin/ that are NOT kept there between rounds (tools/bin is pruned at round end so that only the product
# ships to the GPU box): libmi355rt_stamped.so (`make -C cuda-ray-tracer_amd all STAMPS=1 SPILLS_OK=1`, copy, then `make all` again for the product)
# and, for the A/B against an older round, libmi355rt_rN.so built from that round's tag.  Steps whose library is missing fail and are skipped over.
O=gpurun_out/${1:-r3/final6}
mkdir -p $O
step() { t=$1; out=$2; shift 2; echo "== [$t s] $*"; timeout -k 10 $t bash -c "$*" > $out 2>&1; rc=$?; echo "rc=$rc"; tail -n 2 $out | cut -c1-300; if [ $rc -ge 124 ]; then echo "step killed: stopping"; exit $rc; fi; }
if [ "${PART:-1}" = 1 ]; then
step 900 $O/pytest_gpu.txt "python -m pytest tests -q -m gpu"
step 300 $O/bench.json "python bench.py 2> $O/bench.err"
step 300 $O/bench_force_dist.json "python bench.py --force-dist --steps 50 --warmup 10 --no-cpu-baseline 2> $O/bench_force_dist.err"
step 300 $O/all_configs.txt "python tests/tools/config_bench.py"
step 300 $O/flythrough.txt "python tools/flythrough_bench.py"
step 400 $O/ab_lean_general.txt "MI355RT_LEAN=always python tools/ab_flags.py lean=0 general=256"
step 300 $O/ab_adaptive.txt "python tools/ab_flags.py adaptive=0"
step 300 $O/ab_r2.txt "MI355RT_LIB=tools/bin/libmi355rt_r2.so python tools/ab_flags.py round2=0"
step 200 $O/timeline.txt "TIMELINE_LEAN_NAMES=1 MI355RT_LEAN=always MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/timeline.py"
step 200 $O/timeline_general.txt "MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/timeline.py 1920 1080 256"
step 200 $O/phase_shares.txt "for s in '1920 1080' '3840 2160' '7680 4320'; do MI355RT_LEAN=always MI355RT_LIB=tools/bin/libmi355rt_stamped.so python tools/phase_profile.py 20spheres \$s; done"
fi
if [ "${PART:-1}" = 2 ]; then
step 300 $O/kt.log "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-orbit --no-configs > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json"
step 60 $O/kernel_stats.json "python tools/kernel_stats_summarize.py $O/kt $O/kernel_stats_summary.json"
step 600 $O/pmc.log "bash tools/pmc_profile.sh $O/pmc"
step 60 $O/pmc_summary.txt "python tools/pmc_summarize.py $O/pmc"
step 300 $O/kt4.log "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload config4 --steps 50 --warmup 5 --no-cpu-baseline --no-orbit --no-configs > $GRAFT_REPO_ROOT/$O/bench_config4_under_rocprof.json"
step 600 $O/pmc4.log "BENCH_ARGS='--workload config4' bash tools/pmc_profile.sh $O/pmc4"
step 60 $O/pmc4_summary.txt "BENCH_ARGS='--workload config4' python tools/pmc_summarize.py $O/pmc4 'wavefront_tile_kernel<false'"
step 60 $O/cleanup.txt "find $O/pmc $O/pmc4 $O/kt $O/kt4 -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete"
fi
if [ "${PART:-1}" = 3 ]; then
step 900 $O/fuzz_spheres.txt "python tests/tools/fuzz_spheres.py ${FUZZ_N:-4000} 20000"
step 900 $O/fuzz_parity.txt "python tests/tools/fuzz_parity.py ${FUZZ_N:-4000} 20000"
step 600 $O/fuzz_cubic.txt "python tests/tools/fuzz_cubic.py 1000 20000"
fi
