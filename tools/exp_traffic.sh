# HBM traffic per kernel (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-bunny_1080p}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python3 tools/one_frame.py $W 3 > gpurun_out/pmc_$c.log 2>&1 || { tail -5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 tools/traffic_table.py $W
