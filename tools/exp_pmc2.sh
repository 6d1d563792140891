# second set of counter passes: instruction fetch / branches / scalar cache
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-bunny_1080p}
run() {
  n=$1; shift
  rm -rf gpurun_out/pmc_$n
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$n -o p -- python3 tools/one_frame.py $W 2 > gpurun_out/pmc_$n.log 2>&1 || { tail -5 gpurun_out/pmc_$n.log; exit 1; }
}
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d
run a SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INSTS SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES &&
run b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES &&
run c SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL &&
python3 tools/pmc_table.py
