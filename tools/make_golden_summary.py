"""Fixture from the reference's own mission summary log (log/summary_LSC_10agents.csv, written by
MultiSyncSimulator::saveSummarizedResultAsCSV, reference src/multi_sync_simulator.cpp:658-709): the description line and the
first two mission rows as written -- the format check of shim/include/result_csv.hpp's SimulationSummaryCsv.  Data only; run in
the build container.

    python tools/make_golden_summary.py  ->  tests/golden/summary_log_lines.json
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lines = open("/root/reference/log/summary_LSC_10agents.csv").read().splitlines()[:3]
    out = {"source": "reference log/summary_LSC_10agents.csv, lines 1-3 (description + two missions)", "raw_lines": lines}
    with open(os.path.join(ROOT, "tests", "golden", "summary_log_lines.json"), "w") as f:
        json.dump(out, f)
    print(len(lines), "lines,", len(lines[0].split(",")), "columns")


if __name__ == "__main__":
    main()
