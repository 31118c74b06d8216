"""gpurun_out/pmc<TAG>{1..5} (scripts/pmc_passes.sh) -> profiles JSON: per-kernel counters, HBM bytes per segment, VALU busy.
usage: python scripts/pmc_to_json.py <TAG> <segments> <out.json>"""
import csv, glob, json, sys, collections
tag, segments, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
def short(n):
    for k in ("k_extend", "k_shade", "k_generate", "k_resolve"):
        if k in n:
            return k
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for i in range(1, 6):
    for f in glob.glob(f"gpurun_out/pmc{tag}{i}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            acc[k][r["Counter_Name"] + ("" if r["Counter_Name"] != "GRBM_GUI_ACTIVE" else f"@pass{i}")] += float(r["Counter_Value"])
            if i == 1:
                nd[k].add(r["Dispatch_Id"])
res = {"command": "scripts/pmc_passes.sh: five `rocprofv3 --pmc <group> -- python3 scripts/gpu_render_once.py book1 1200 800 500 1` passes (one render each; "
                  "SQ groups, FETCH_SIZE and WRITE_SIZE in separate passes)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests of 16 B/lane reads at 64 B, MI355X_MICROARCH.md); "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = VALU busy",
       "segments": segments, "kernels": {}}
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE@pass1", 0.0) / 8.0
    e = {"dispatches": len(nd[k]), "counters": {n: v for n, v in sorted(c.items())}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        e.update(hbm_read_bytes=rd, hbm_write_bytes=wr, read_bytes_per_segment=round(rd / segments, 2), write_bytes_per_segment=round(wr / segments, 2),
                 bytes_per_segment=round((rd + wr) / segments, 2))
    if g > 0 and "SQ_INSTS_VALU" in c:
        e.update(gpu_cycles=g, valu_busy=round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / g, 3), valu_lane_utilisation=round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), 3),
                 valu_wave_instructions_per_segment=round(c["SQ_INSTS_VALU"] / segments, 2))
    if "SQ_LDS_IDX_ACTIVE" in c:
        e.update(lds_conflict_share=round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 3))
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(k, {x: e[x] for x in e if x != "counters"})
