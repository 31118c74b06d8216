# On the GPU box: the evidence committed under profiles/ for this round — rocprofv3 kernel-trace stats of bench.py, the PMC passes of the
# bench workload (book-1, LDS-resident), of the config-5 scene (compressed records in HBM) and of configs 3 and 4 (book-2 final scene, Cornell box), every BASELINE config, the three walks
# of config 5, and one plain bench line. Everything lands in gpurun_out/prof3/ (copied to profiles/ afterwards, on the CPU side,
# by scripts/collect_profiles.py).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3/bench_stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-variants > gpurun_out/prof3/bench_under_rocprof.json 2> gpurun_out/prof3/bench_under_rocprof.err || exit 1
echo "kernel-trace done" >> gpurun_out/prof3/progress.log
bash scripts/pmc_passes.sh R3 > gpurun_out/prof3/pmc_book1_summary.txt 2>&1 || exit 1
echo "pmc book1 done" >> gpurun_out/prof3/progress.log
bash scripts/pmc_c5.sh C5c 16 > gpurun_out/prof3/pmc_c5_summary.txt 2>&1 || exit 1
echo "pmc c5 done" >> gpurun_out/prof3/progress.log
bash scripts/pmc_scene.sh C3r final 800 800 200 || exit 1
echo "pmc c3 done" >> gpurun_out/prof3/progress.log
bash scripts/pmc_scene.sh C4r cornell 600 600 500 || exit 1
echo "pmc c4 done" >> gpurun_out/prof3/progress.log
# every BASELINE config under the kernel trace: the per-kernel durations the C3/C4/C5 lines of bench.py's `other_configs` can be checked against
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3/configs_stats -o run -- python3 scripts/gpu_configs.py > gpurun_out/prof3/configs.log 2>&1 || exit 1
cp gpurun_out/configs.json gpurun_out/prof3/configs.json
timeout -k 10 300 python3 scripts/gpu_c5_walks.py 16 > gpurun_out/prof3/c5_walks.log 2>&1 || exit 1
echo "configs done" >> gpurun_out/prof3/progress.log
# the PMC pass of the bench workload -> profiles/r03_pmc_book1.json of THIS copy, so that the bench line below carries the roofline of the
# sources it runs (scripts/collect_profiles.py repeats the conversion on the CPU side, where git is, for the committed file)
SEG=$(grep -o "segments [0-9]*" gpurun_out/pmcR31.log | tail -1 | cut -d" " -f2)
python3 scripts/pmc_to_json.py R3 $SEG profiles/r03_pmc_book1.json "scripts/pmc_passes.sh R3 (on the GPU box)" > /dev/null || exit 1
SEG5=$(grep -o "segments [0-9]*" gpurun_out/pmcC5c1.log | tail -1 | cut -d" " -f2)
python3 scripts/pmc_to_json.py C5c $SEG5 profiles/r03_pmc_c5.json "scripts/pmc_c5.sh C5c 16 (on the GPU box)" > /dev/null || exit 1
timeout -k 10 500 python3 bench.py > gpurun_out/prof3/bench.json 2> gpurun_out/prof3/bench.err || exit 1
cat gpurun_out/prof3/bench.json
