# On the GPU box: the evidence committed under profiles/ (kernel-trace stats of bench.py, PMC passes, all configs).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/bench_stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/prof/bench_under_rocprof.log 2>&1 || exit 1
bash scripts/pmc_passes.sh R > gpurun_out/prof/pmc_summary.txt 2>&1 || exit 1
timeout -k 10 400 python3 scripts/gpu_configs.py > gpurun_out/prof/configs.log 2>&1 || exit 1
cp gpurun_out/configs.json gpurun_out/prof/configs.json
timeout -k 10 500 python3 bench.py > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err || exit 1
cat gpurun_out/prof/bench.json
