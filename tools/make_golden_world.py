"""Fixture: the obstacle boxes of the reference's forest10 world (world/forest/forest10.csv: centre x,y,z, size x,y,z per
row, read by MapManager::updateOctreeFromCSV, reference src/map_manager.cpp:262-305), the start points of
missions/forest10/forest10_10.json and the world dimension, plus the one corridor face the reference's result log pins:
agent 1's -x face at 2.55 (the value with which the logged trajectory reproduces to its printed
digits, tests/golden/kat_log.json).  Data only; run in the build container.

    python tools/make_golden_world.py  ->  tests/golden/forest10_world.json
"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    boxes = [[float(v) for v in row[:6]] for row in csv.reader(open(os.path.join(REF, "world/forest/forest10.csv"))) if len(row) >= 6]
    m = json.load(open(os.path.join(REF, "missions/forest10/forest10_10.json")))
    dim = m["world"][0]["dimension"]
    out = {"source": "reference world/forest/forest10.csv, missions/forest10/forest10_10.json; launch/simulation.launch:56-60 (2-D, z = 0.6, "
                     "resolution 0.1); pinned face from log/simulation_1663743693.650981_LSC_10agents.csv via SURVEY.md 8c",
           "boxes": boxes, "world_min": dim[:3], "world_max": dim[3:], "resolution": 0.1, "max_dist": 1.0, "z_2d": 0.6, "radius": 0.15,
           "starts": [a["start"][:2] + [0.6] for a in m["agents"]], "goals": [a["goal"][:2] + [0.6] for a in m["agents"]],
           "pinned": {"agent": 1, "face": "bmin_x", "value": 2.55}}
    with open(os.path.join(ROOT, "tests", "golden", "forest10_world.json"), "w") as f:
        json.dump(out, f)
    print(len(boxes), "boxes,", len(out["starts"]), "agents")


if __name__ == "__main__":
    main()
