# usage (on the GPU box): bash scripts/pmc_lds_layouts.sh — the LDS counters of k_extend on the bench workload for the three layouts of the LDS-resident
# records (RT_LDS_RECORDS = 0: 32-byte records, 1: halves apart, 2: 48 bytes apart); one rocprofv3 --pmc pass each.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in 0 1 2; do
  export RT_LDS_RECORDS=$L
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcLL$L -o run -- python3 scripts/gpu_render_once.py book1 1200 800 500 1 > gpurun_out/pmcLL$L.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmcLL$L/**/*counter_collection.csv", recursive=True)[0]
s = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f)):
    if "k_extend" in r["Kernel_Name"]:
        s[r["Counter_Name"]] += float(r["Counter_Value"])
print("layout $L:", {k: "%.4g" % v for k, v in sorted(s.items())}, "conflict share %.3f" % (s["SQ_LDS_BANK_CONFLICT"] / max(1.0, s["SQ_LDS_IDX_ACTIVE"])), "array cycles per LDS instruction %.2f" % (s["SQ_LDS_IDX_ACTIVE"] / max(1.0, s["SQ_INSTS_LDS"])))
PY
done
