# On the GPU box: vector-memory-side counters of the config-5 walk (TA / TCP / address translation), two passes. usage: bash scripts/pmc_c5_mem.sh [spp]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SPP=${1:-16}
for i in 1 2 3; do
  case $i in
    1) C="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE";;
    2) C="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE";;
    3) C="TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE";;
  esac
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmcMEM$i -o run -- python3 scripts/gpu_c5.py $SPP > gpurun_out/pmcMEM$i.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for i in (1, 2, 3):
    acc = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmcMEM{i}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_extend" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f"pass {i} (k_extend, all dispatches):", {k: v for k, v in acc.items()})
PY
