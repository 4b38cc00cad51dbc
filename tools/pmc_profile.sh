#!/bin/bash
# PMC counter passes for the render kernel of `bench.py $BENCH_ARGS` (plain launches: --no-graph; the counters are per dispatch either way).
# One counter group per rocprofv3 run; --pmc is never combined with trace domains other than kernel-trace.
# Usage on the GPU box: [BENCH_ARGS="--workload config4"] tools/pmc_profile.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$OUT/$name" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-orbit --no-configs --no-graph $BENCH_ARGS > "$R/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run sq4 SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
