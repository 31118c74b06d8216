cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o run -- python3 scripts/gpu_render_once.py book1 1200 800 500 2 > gpurun_out/kt.log 2>&1
ls gpurun_out/kt
