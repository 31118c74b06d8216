cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt5 -o run -- python3 scripts/gpu_c5.py 16 > gpurun_out/kt5.log 2>&1
tail -2 gpurun_out/kt5.log
python3 scripts/trace_tail.py gpurun_out/kt5/run_kernel_trace.csv
