# rocprofv3 kernel trace + stats of frames whose launches run one at a time (tools/one_frame.py ... serial): the per-kernel
# averages to hold against bench.py's HIP-event durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_serial
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o serial -- python3 tools/one_frame.py ${1:-bunny_1080p} 20 serial > gpurun_out/prof_serial.txt 2> gpurun_out/prof_serial.err
