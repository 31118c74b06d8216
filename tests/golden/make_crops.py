#!/usr/bin/env python3
"""Writes tests/golden/crops_<config>.npz: the f64 ORACLE's rgb sums (and traversal counters) on the crops of the
benchmarked frames listed in tests/crops.py. CPU only; run from the repo root:

    python tests/golden/make_crops.py            # all configs
    python tests/golden/make_crops.py C2 C4      # some

The fixtures are the oracle's output (the reference itself cannot be run: SURVEY F1-F3), so they pin the GPU path to the
oracle at the sizes that are benchmarked; tests/test_full_frame_fixtures.py re-derives one crop per config on the CPU so
that a change of the oracle cannot go unnoticed."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import crops as K          # noqa: E402
import rta                 # noqa: E402
from oracle import binding as orc   # noqa: E402


def oracle_crops(pkg, name, tmp_dir, n_threads, earth=None, only=None):
    cfg = K.CONFIGS[name]
    hs = K.host_scene(pkg, name, tmp_dir, earth=earth)
    cam = hs.camera(cfg["width"] / cfg["height"])
    prm = pkg.make_params(cfg["width"], cfg["height"], cfg["spp"], max_depth=50, seed=cfg["seed"])
    names = [n for n in cfg["crops"] if only is None or n in only]
    imgs, stats = orc.render_crops(hs.desc, cam, prm, [cfg["crops"][n] for n in names], precision=64, n_threads=n_threads, count=True)
    # per-pixel standard error of the mean (per channel), from the oracle's own per-sample radiances: the scale SURVEY 8(d)'s converged-mean
    # test measures a difference against (|d| <= 4 SE). Needs enough samples per pixel to mean anything: skipped below 64 spp (config 5).
    ses = []
    for n in names:
        if cfg["spp"] < 64:
            ses.append(None)
            continue
        _, _, ps = orc.render(hs.desc, cam, prm, precision=64, n_threads=n_threads, rect=cfg["crops"][n], per_sample=True)
        ses.append((ps.std(axis=2, ddof=1) / np.sqrt(cfg["spp"])).astype(np.float32))
        del ps
    return names, imgs, stats, ses


def main():
    pkg = rta.load()
    orc.build()
    which = sys.argv[1:] or list(K.CONFIGS)
    earth = None
    nt = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as tmp:
        for name in which:
            cfg = K.CONFIGS[name]
            if cfg.get("earth") and earth is None:
                from PIL import Image
                earth = np.asarray(Image.open(os.path.join(K.GOLDEN, "earthmap_rgb.png")).convert("RGB"))
            t0 = time.time()
            names, imgs, stats, ses = oracle_crops(pkg, name, tmp, nt, earth)
            out = {}
            for n, img, st, se in zip(names, imgs, stats, ses):
                out[n] = img
                out[n + "__counters"] = np.array([st["samples"], st["segments"], st["node_tests"]] + st["prim_tests"], dtype=np.uint64)
                if se is not None:
                    out[n + "__se"] = se
            np.savez_compressed(K.golden_path(name), **out)
            print(f"{name}: {len(names)} crops, {sum(s['samples'] for s in stats)} samples, {time.time() - t0:.1f} s ->", K.golden_path(name),
                  os.path.getsize(K.golden_path(name)) // 1024, "KiB")


if __name__ == "__main__":
    main()
