"""Fixture from the reference's own result log: the positions of the 10 agents of the forest10 run at every logged time
(log/simulation_1663743693.650981_LSC_10agents.csv, written by MultiSyncSimulator::saveSimulationResultAsCSV,
reference src/multi_sync_simulator.cpp:586-656) and the summary figures of the same run (log/summary_LSC_10agents.csv,
row start_time = 1663743693.650981: safety_ratio_agent, vel / acc excess ratios).  Data only; run in the build container.

    python tools/make_golden_simlog.py  ->  tests/golden/sim_log_states.json
"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/log"


def main():
    path = os.path.join(REF, "simulation_1663743693.650981_LSC_10agents.csv")
    rows = list(csv.reader(open(path)))[1:]
    raw = open(path).read().splitlines()[:4]  # header + the first three rows as written (format check of the CSV writer)
    t, pos, vel, acc, ptime = [], [], [], [], []
    for row in rows:
        t.append(float(row[1]))
        pos.append([[float(v) for v in row[12 * q + 2:12 * q + 5]] for q in range(10)])
        vel.append([[float(v) for v in row[12 * q + 5:12 * q + 8]] for q in range(10)])
        acc.append([[float(v) for v in row[12 * q + 8:12 * q + 11]] for q in range(10)])
        ptime.append([float(row[12 * q + 11]) for q in range(10)])
    summ = [r for r in csv.DictReader(open(os.path.join(REF, "summary_LSC_10agents.csv"))) if r["start_time"] == "1663743693.650981"][0]
    out = {"source": "reference log/simulation_1663743693.650981_LSC_10agents.csv (columns id,t,px..az per agent) and "
                     "log/summary_LSC_10agents.csv (same start_time)",
           "radius": 0.15, "downwash": 2.0, "vmax": 1.0, "amax": 2.0, "t": t, "pos": pos, "vel": vel, "acc": acc,
           "planning_time": ptime[:3], "raw_lines": raw,
           "summary": {k: float(summ[k]) for k in ("safety_ratio_agent", "vel_excess_ratio", "acc_excess_ratio", "total_flight_time")}}
    with open(os.path.join(ROOT, "tests", "golden", "sim_log_states.json"), "w") as f:
        json.dump(out, f)
    print(len(t), "rows", out["summary"])


if __name__ == "__main__":
    main()
