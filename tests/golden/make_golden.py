"""Generates the golden framebuffers under tests/golden/ with the repo's own f64 CPU oracle
(the reference cannot be run: no Rust toolchain, and its RNG is unseedable — SURVEY.md F1/F2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rta  # noqa: E402
from oracle import binding as orc  # noqa: E402

pkg = rta.load()
here = os.path.dirname(os.path.abspath(__file__))
hs = pkg.HostScene("book1", 1)
img, _ = orc.render(hs.desc, hs.camera(64 / 40), pkg.make_params(64, 40, 8, seed=1), precision=64, n_threads=8)
np.save(os.path.join(here, "book1_64x40_8spp_f64.npy"), img)
print("book1", img.mean())
