#!/bin/bash
# Counter passes for one workload (run on the GPU box: gpurun -- bash tools/collect_counters.sh <workload>).
# Every pass is a rocprofv3 run of its own (the guide's rule: --pmc alone, never with a trace domain; at most 8 SQ /
# 4 TCC counters per pass, FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), plus one --kernel-trace --stats run for the
# durations. tools/counters_table.py turns the CSVs into gpurun_out/counters_<workload>.json (copy to profiles/rN/).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-bunny_1080p}
LIM=${2:-240}
pass() {   # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/ctr_$name
  timeout -s KILL $LIM rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/ctr_$name -o p -- python3 tools/frames_for_counters.py $W > gpurun_out/ctr_$name.log 2>&1 \
    || { echo "pass $name failed"; tail -5 gpurun_out/ctr_$name.log; exit 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
pass tcc TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
rm -rf gpurun_out/ctr_trace
timeout -k 10 $LIM rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctr_trace -o p -- python3 tools/frames_for_counters.py $W > gpurun_out/ctr_trace.log 2>&1 \
  || { echo "trace pass failed"; tail -5 gpurun_out/ctr_trace.log; exit 1; }
python3 tools/counters_table.py $W
# the raw CSVs of a large frame are hundreds of MB (gpurun_out/ is copied back only up to 64 MiB): keep the table's JSON
rm -rf gpurun_out/ctr_sq1 gpurun_out/ctr_sq2 gpurun_out/ctr_fetch gpurun_out/ctr_write gpurun_out/ctr_tcc
mkdir -p gpurun_out/trace_$W && cp gpurun_out/ctr_trace/*kernel_stats.csv gpurun_out/trace_$W/ 2>/dev/null; rm -rf gpurun_out/ctr_trace
