"""Build tuning variants here (CPU), then on the GPU box run each with bench-like timing.
usage: python scripts/sweep.py build | run"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {
    "base": [],
    "sh1024": ["RT_SHADE_THREADS=1024"],
    "sh256": ["RT_SHADE_THREADS=256"],
    "dbgwork": ["RT_DEBUG_WORK=3000u"],
    "b1r1s1": ["RT_LEAF_BATCH=1", "RT_REFILL_MIN=1", "RT_STEPS=1"],
    "b64": ["RT_LEAF_BATCH=64"],
    "f64only": ["RT_SPHERE_F64_ONLY=1"],
    "tol6": ["RT_SPHERE_TOL=1e-6f"],
    "tol4": ["RT_SPHERE_TOL=1e-4f"],
    "libm": ["RT_LIBM_SINCOS=1"],
    "stamps": ["RT_STAMPS=1"],
    "sk9": ["RT_SERVE_KINDS_MIN=9u"], "sk9b32": ["RT_SERVE_KINDS_MIN=9u", "RT_LEAF_BATCH=32"], "sk9b16": ["RT_SERVE_KINDS_MIN=9u", "RT_LEAF_BATCH=16"],
    "c16all": ["RT_C16_LOAD_ALL=1"],
    "sw7": ["RT_SHADE_WAVES=7"], "sw5": ["RT_SHADE_WAVES=5"], "sw6": ["RT_SHADE_WAVES=6"], "sw8": ["RT_SHADE_WAVES=8"],
    "sb16": ["RT_SERVE_BEST=16"], "sb24": ["RT_SERVE_BEST=24"], "sb32": ["RT_SERVE_BEST=32"], "sb40": ["RT_SERVE_BEST=40"], "sb48": ["RT_SERVE_BEST=48"],
    "b32r24": ["RT_LEAF_BATCH=32"], "b40": ["RT_LEAF_BATCH=40"], "b48": ["RT_LEAF_BATCH=48"],
    "r1": ["RT_REFILL_MIN=1"],
    "r8": ["RT_REFILL_MIN=8"],
    "r24": ["RT_REFILL_MIN=24"],
    "r32": ["RT_REFILL_MIN=32"],
    "r48": ["RT_REFILL_MIN=48"],
    "t512": ["RT_EXTEND_THREADS=512"],
    "t1024": ["RT_EXTEND_THREADS=1024"],
    "c128": ["RT_CHUNK=128"],
    "longwalk": ["RT_DEBUG_LONGWALK=1"],
    "c256": ["RT_CHUNK=256"],
    "b16r16": ["RT_LEAF_BATCH=16", "RT_REFILL_MIN=16"],
    "c512": ["RT_CHUNK=512"],
    "c1024": ["RT_CHUNK=1024"],
    "c2048": ["RT_CHUNK=2048"],
    "c4096": ["RT_CHUNK=4096"],
    "c8192": ["RT_CHUNK=8192"],
    "c2048b24r24": ["RT_CHUNK=2048", "RT_LEAF_BATCH=24", "RT_REFILL_MIN=24"],
    "s2": ["RT_STEPS=2"],
    "s4": ["RT_STEPS=4"],
    "s8": ["RT_STEPS=8"],
    "s6": ["RT_STEPS=6"],
    "s3": ["RT_STEPS=3"],
    "s5": ["RT_STEPS=5"],
    "b8": ["RT_LEAF_BATCH=8"],
    "b24": ["RT_LEAF_BATCH=24"],
    "b32": ["RT_LEAF_BATCH=32"],
    "b1": ["RT_LEAF_BATCH=1"],
    "bs0": ["RT_BLOCK_SHIFT=0"],
    "bs1": ["RT_BLOCK_SHIFT=1"],
    "bs2": ["RT_BLOCK_SHIFT=2"],
    "bs3": ["RT_BLOCK_SHIFT=3"],
    "bs5": ["RT_BLOCK_SHIFT=5"],
    "o6": ["RT_EXTEND_PER_CU_MAX=6"],
    "o5": ["RT_EXTEND_PER_CU_MAX=5"],
    "o4": ["RT_EXTEND_PER_CU_MAX=4"],
    "o3": ["RT_EXTEND_PER_CU_MAX=3"],
    "o2": ["RT_EXTEND_PER_CU_MAX=2"],
}
if sys.argv[1] == "build":
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "ray-tracer-archive_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    from concurrent.futures import ThreadPoolExecutor
    names = sys.argv[2:] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        for n, p in zip(names, ex.map(lambda n: b.build_variant(n, VARIANTS[n]), names)):
            print(n, p)
else:
    names = sys.argv[2:] or list(VARIANTS)
    pools = [int(x) for x in os.environ.get("POOLS", "0").split(",")]
    for n in names:
        for pool in pools:
            env = dict(os.environ, RT_HIP_LIB=os.path.join(ROOT, "ray-tracer-archive_amd", "lib", "variants", f"librt_hip_{n}.so"))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--spp", os.environ.get("SPP", "200"),
                                "--pool-slots", str(pool), "--no-variants"], env=env, capture_output=True, text=True)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{n:8s} pool {pool:9d}: {j['value']:8.1f} Msamples/s  extend {j['kernel_ms_per_step']['k_extend']:7.1f} ms shade {j['kernel_ms_per_step']['k_shade']:6.1f} ms other {j['kernel_ms_per_step']['generate+resolve']:5.1f} ms  step {j['ms_per_step']:7.1f} ms", flush=True)
            except Exception as e:
                print(n, pool, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)
