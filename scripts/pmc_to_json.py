"""gpurun_out/pmc<TAG>{1..5} (scripts/pmc_passes.sh or pmc_c5.sh) -> profiles JSON: per-kernel counters and the derived figures
bench.py and DESIGN.md quote. Run HERE (the CPU container) after the GPU call has merged gpurun_out/.

usage: python scripts/pmc_to_json.py <TAG> <segments> <out.json> ["command text"]

Units (MI355X_MICROARCH.md): a wave64 VALU instruction takes 2 cycles on a SIMD-32 (>= 2 waves per SIMD); SQ_THREAD_CYCLES_VALU = active
lanes summed over VALU instructions (lane-instructions); GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB and
FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B-per-lane reads at 64 B). A utilisation above 1 is a bug in this
script, not a result: it aborts."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

tag, segments, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
command = sys.argv[4] if len(sys.argv) > 4 else "scripts/pmc_passes.sh"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CU, N_SIMD, LANES_PER_CLK = 256, 1024, 32768


def short(n):
    for k in ("k_extend", "k_shade", "k_generate", "k_resolve", "k_drain"):
        if k in n:
            return k
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
cyc = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc{tag}[0-9]"))):
    per_pass = collections.defaultdict(float)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                per_pass[k] += float(r["Counter_Value"])
            else:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if d.endswith("1"):
                nd[k].add(r["Dispatch_Id"])
    for k, v in per_pass.items():
        cyc[k].append(v / 8.0)

try:
    src_hash = open(os.path.join(ROOT, f"gpurun_out/pmc{tag}_source_hash.txt")).read().strip()
except OSError:
    src_hash = None
try:
    git = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    git = None
res = {"command": command, "source_hash": src_hash, "git_commit": git, "segments": segments,
       "units": "FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950); GRBM_GUI_ACTIVE / 8 = GPU cycles (mean over the SQ passes); wave64 VALU = 2 cycles "
                "on a SIMD-32; SQ_THREAD_CYCLES_VALU = lane-instructions; lane peak = 32768 per cycle; LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles",
       "kernels": {}}
for k, c in acc.items():
    g = sum(cyc[k]) / len(cyc[k]) if cyc[k] else 0.0
    e = {"dispatches": len(nd[k]), "gpu_cycles": g, "counters": {n: v for n, v in sorted(c.items())}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        e.update(hbm_read_bytes=rd, hbm_write_bytes=wr, read_bytes_per_segment=round(rd / segments, 2), write_bytes_per_segment=round(wr / segments, 2),
                 bytes_per_segment=round((rd + wr) / segments, 2))
    if g > 0 and "SQ_INSTS_VALU" in c:
        e.update(valu_lane_instructions_per_segment=round(c["SQ_THREAD_CYCLES_VALU"] / segments, 1),
                 valu_wave_instructions_per_segment=round(c["SQ_INSTS_VALU"] / segments, 2),
                 valu_lane_frac=round(c["SQ_THREAD_CYCLES_VALU"] / (g * LANES_PER_CLK), 4),
                 valu_issue_frac=round(c["SQ_INSTS_VALU"] * 2 / (g * N_SIMD), 4),
                 valu_lane_utilisation=round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64), 4))
        if "SQ_WAVE_CYCLES" in c:
            e.update(wait_inst_any_share=round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 4), wait_any_share=round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 4))
        for name in ("valu_lane_frac", "valu_issue_frac", "valu_lane_utilisation"):
            assert e[name] <= 1.0, (k, name, e[name], "a utilisation above 1 is a unit error")
    if g > 0 and "SQ_LDS_IDX_ACTIVE" in c:
        e.update(lds_busy=round(c["SQ_LDS_IDX_ACTIVE"] / N_CU / g, 4), lds_conflict_share=round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 4))
        assert e["lds_busy"] <= 1.0, (k, e["lds_busy"])
    if "SQ_INSTS_SALU" in c and "SQ_INSTS_VALU" in c:
        e.update(salu_per_valu=round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3))
    if "TCC_HIT_sum" in c:
        e.update(l2_hit_rate=round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4))
    if g > 0 and "hbm_read_bytes" in e:
        # seconds of the kernel at the clock the SQ passes ran at cannot be had from counters alone: GB per GPU-cycle x 2.4e9 is an upper bound
        e.update(hbm_bytes_per_gpu_cycle=round((e["hbm_read_bytes"] + e["hbm_write_bytes"]) / g, 2),
                 hbm_frac_of_8tbs_at_2p4ghz=round((e["hbm_read_bytes"] + e["hbm_write_bytes"]) / g * 2.4e9 / 8e12, 4))
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(k, {x: e[x] for x in e if x != "counters"})
