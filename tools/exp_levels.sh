# kernel timeline of the last frame of a short bench run (start offset and duration in us, per kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_lv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lv -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ${@} > gpurun_out/lv.json 2> gpurun_out/lv.err || exit 1
python3 tools/timeline.py
