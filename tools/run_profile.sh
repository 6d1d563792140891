# rocprofv3 kernel trace + stats of the default bench command; summaries go to gpurun_out/prof_final (copy to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_final
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py ${@} > gpurun_out/prof_final_bench.json 2> gpurun_out/prof_final_bench.err
