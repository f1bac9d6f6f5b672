"""Golden vectors for the hull closest-point routine, produced by the REFERENCE's own openGJK.

Run in the build container only (needs /root/reference): compiles oracle/_ref/libref_gjk.so from the reference's
src/openGJK/openGJK.cpp (oracle/Makefile, target `ref`) and records its outputs on seeded 6-point hulls of the kinds the
LSC generation meets (generic, planar = 2-D missions, repeated points = hovering agents, float32-rounded coordinates,
origin inside the hull).  The fixture is data only: inputs and the reference's outputs.

    python tools/make_golden_gjk.py  ->  tests/golden/gjk_hulls.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

assert O.build_ref(), "needs the reference checkout to build oracle/_ref"
rng = np.random.default_rng(20240928)
cases = []
for t in range(240):
    c = rng.normal(size=3) * rng.uniform(0.2, 3.0)
    pts = c + rng.normal(size=(6, 3)) * rng.uniform(0.01, 1.0)
    kind = "generic"
    if t % 5 == 0:
        pts[:, 2] = 0.0
        kind = "planar"
    if t % 7 == 0:
        pts[3:] = pts[:3]
        kind += "+repeated"
    if t % 3 == 0:
        pts = pts.astype(np.float32).astype(np.float64)
        kind += "+f32"
    if t % 13 == 0:
        pts = pts - pts.mean(0) * rng.uniform(0.8, 1.2)  # hull around the origin
        kind += "+around-origin"
    d, v = O.ref_gjk(pts)
    cases.append({"kind": kind, "hull": pts.tolist(), "dist": d, "closest": v.tolist()})
out = {"source": "reference src/openGJK/openGJK.cpp (gjk) called as in include/geometry.hpp:266-296, query point = origin",
       "cases": cases}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "gjk_hulls.json"), "w"))
print(len(cases), "cases;", sum(c["dist"] == 0 for c in cases), "with the origin inside the hull")
