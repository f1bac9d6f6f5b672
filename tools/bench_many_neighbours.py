import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import bench_generic as B
from lsc_dr_planner_amd import api
for N, M, dim, n_obs, mode, name in ((128, 5, 3, 64, api.PLANNER_LSC, "lsc"), (128, 5, 3, 64, api.PLANNER_DLSC, "dlsc"), (64, 5, 3, 48, api.PLANNER_LSC, "lsc"), (1024, 5, 3, 48, api.PLANNER_LSC, "lsc"),
                                     (64, 5, 3, 40, api.PLANNER_DLSC, "dlsc"), (64, 5, 3, 100, api.PLANNER_LSC, "lsc")):
    out = []
    for pin, force in ((None, False), ("2", False), ("4", False), (None, True)):
        os.environ.pop("LSCQP_WAVES", None)
        if pin: os.environ["LSCQP_WAVES"] = pin
        try:
            r = B.run(N, M, dim, n_obs, mode, force)
            out.append("%s%s: %.3f ms it %.2f/%d bad %d" % ("generic" if force else "selected", " W=" + pin if pin else "", r[0], r[1], r[2], r[3]))
        except Exception as ex:
            out.append("%s: %s" % (pin, str(ex)[:40]))
    os.environ.pop("LSCQP_WAVES", None)
    print(N, M, dim, n_obs, name, " | ".join(out))
