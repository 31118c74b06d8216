"""After `gpurun -- bash scripts/profile_round.sh`: turn gpurun_out/prof3 + the PMC pass directories into the files kept under profiles/
(run here, on the CPU side, where git is). usage: python scripts/collect_profiles.py"""
import glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
def seg_of(log):
    m = re.findall(r"segments (\d+)", open(log).read())
    return int(m[-1])
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), "R3", str(seg_of(os.path.join(G, "pmcR31.log"))), os.path.join(P, "r03_pmc_book1.json"),
                       "scripts/pmc_passes.sh R3: five rocprofv3 --pmc passes of python3 scripts/gpu_render_once.py book1 1200 800 500 1 (the bench workload, one render each)"])
subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), "C5c", str(seg_of(os.path.join(G, "pmcC5c1.log"))), os.path.join(P, "r03_pmc_c5.json"),
                       "scripts/pmc_c5.sh C5c 16: four rocprofv3 --pmc passes of python3 scripts/gpu_c5.py 16 (1 M spheres + 262 K triangles, SAH tree, 16-byte records in HBM, "
                       "2048x2048x16, two renders per pass; FETCH_SIZE and WRITE_SIZE in passes of their own)"])
for tag, scene, out in (("C3r", "final 800 800 200", "r03_pmc_c3.json"), ("C4r", "cornell 600 600 500", "r03_pmc_c4.json")):
    if os.path.exists(os.path.join(G, f"pmc{tag}1.log")):
        subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), tag, str(seg_of(os.path.join(G, f"pmc{tag}1.log"))), os.path.join(P, out),
                               f"scripts/pmc_scene.sh {tag} {scene}: four rocprofv3 --pmc passes of python3 scripts/gpu_render_once.py {scene} 1 (one render each; FETCH_SIZE and "
                               "WRITE_SIZE in passes of their own)"])
for f in glob.glob(os.path.join(G, "prof3", "bench_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r03_bench_kernel_stats.csv"))
for f in glob.glob(os.path.join(G, "prof3", "configs_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r03_configs_kernel_stats.csv"))
for src, dst in (("bench_under_rocprof.json", "r03_bench_under_rocprof.json"), ("bench.json", "r03_bench.json"), ("configs.json", "r03_configs_1gpu.json"), ("c5_walks.log", "r03_c5_walks.txt")):
    shutil.copy(os.path.join(G, "prof3", src), os.path.join(P, dst))
print(sorted(os.listdir(P)))
