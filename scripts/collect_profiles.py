"""After `gpurun -- bash scripts/profile_round.sh`: turn gpurun_out/prof2 + the PMC pass directories into the files kept under profiles/
(run here, on the CPU side, where git is). usage: python scripts/collect_profiles.py"""
import glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
def seg_of(log):
    m = re.findall(r"segments (\d+)", open(log).read())
    return int(m[-1])
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), "R2", str(seg_of(os.path.join(G, "pmcR21.log"))), os.path.join(P, "r02_pmc_book1.json"),
                       "scripts/pmc_passes.sh R2: five rocprofv3 --pmc passes of python3 scripts/gpu_render_once.py book1 1200 800 500 1 (the bench workload, one render each)"])
subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), "C5b", str(seg_of(os.path.join(G, "pmcC5b1.log"))), os.path.join(P, "r02_pmc_c5.json"),
                       "scripts/pmc_c5.sh C5b 16: four rocprofv3 --pmc passes of python3 scripts/gpu_c5.py 16 (1 M spheres + 262 K triangles, SAH tree, 16-byte records in HBM, "
                       "2048x2048x16, two renders per pass; FETCH_SIZE and WRITE_SIZE in passes of their own)"])
for tag, scene, out in (("C3", "final 800 800 200", "r02_pmc_c3.json"), ("C4", "cornell 600 600 500", "r02_pmc_c4.json")):
    if os.path.exists(os.path.join(G, f"pmc{tag}1.log")):
        subprocess.check_call([py, os.path.join(ROOT, "scripts", "pmc_to_json.py"), tag, str(seg_of(os.path.join(G, f"pmc{tag}1.log"))), os.path.join(P, out),
                               f"scripts/pmc_scene.sh {tag} {scene}: four rocprofv3 --pmc passes of python3 scripts/gpu_render_once.py {scene} 1 (one render each; FETCH_SIZE and "
                               "WRITE_SIZE in passes of their own)"])
for f in glob.glob(os.path.join(G, "prof2", "bench_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r02_bench_kernel_stats.csv"))
for src, dst in (("bench_under_rocprof.json", "r02_bench_under_rocprof.json"), ("bench.json", "r02_bench.json"), ("configs.json", "r02_configs_1gpu.json"), ("c5_walks.log", "r02_c5_walks.txt")):
    shutil.copy(os.path.join(G, "prof2", src), os.path.join(P, dst))
print(sorted(os.listdir(P)))
