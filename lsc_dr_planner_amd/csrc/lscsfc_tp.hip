// lscsfc_tp.hip -- the THROUGHPUT build of the corridor kernel: lscsfc.hip compiled a second time with 256 threads per agent, the
// registers capped at 128 per lane and the batch tables cut to 35 KB of LDS, so that FOUR agents' workgroups share a CU and their chains
// of dependent tests overlap.  4096 agents (tools/bench_next_rows.py, free-space table in place): 2.28 ms with the latency build
// (1024 threads, one workgroup per CU), 1.39 ms with 512 threads x 2 per CU, 1.06 ms with this one; without the table 2.89 -> 2.01 ms.
// Same statements as the latency build, same boxes bit for bit (tests/test_sfc.py runs both builds; tools/sweep_corridors.py --prepare:
// 36 of 36 world / mode combinations); lscsfc.hip picks the build per launch (agents > CUs and a map of at most 512 cells per axis ->
// this one: its cell-centre table is half as long).  Only lscsfc_launch_throughput_ is exported from this translation unit.
#ifndef LSCSFC_TP_THREADS
#define LSCSFC_TP_THREADS 256
#endif
#ifndef LSCSFC_TP_TODO
#define LSCSFC_TP_TODO 2048
#endif
#ifndef LSCSFC_TP_TAB
#define LSCSFC_TP_TAB 2048
#endif
#ifndef LSCSFC_TP_CELL
#define LSCSFC_TP_CELL 512
#endif
#ifndef LSCSFC_TP_AHEAD
#define LSCSFC_TP_AHEAD 62  // (ONE wavefront assembles the look-ahead: here the instructions count, not the chain -- 4096 agents, INIT / FROM_HULL:
                            // 30 / 46 / 62 / 126 / 254 tests ahead -> 1.00 / 0.99 / 0.97 / 1.01 / 1.08 ms and 1.05 / 1.05 / 1.03 / 1.07 / 1.15 ms)
#endif
#define LSCSFC_AHEAD LSCSFC_TP_AHEAD
#ifdef LSCSFC_TP_SAMPLED
#define LSCSFC_SAMPLED LSCSFC_TP_SAMPLED
#endif
#define LSCSFC_THREADS LSCSFC_TP_THREADS
#define LSCSFC_WAVES_PER_EU 4
#define LSCSFC_TODO LSCSFC_TP_TODO
#define LSCSFC_TAB LSCSFC_TP_TAB
#define LSCSFC_CELL LSCSFC_TP_CELL
#define LSCSFC_VARIANT_ONLY 1
#define lscsfc lscsfc_tp  // the kernels of this build get their own namespace (distinct symbols)
#include "lscsfc.hip"

extern "C" int lscsfc_throughput_max_cells_(void) { return LSCSFC_TP_CELL; }
