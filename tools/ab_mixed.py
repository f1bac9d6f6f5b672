"""A/B on one GPU: fp64 vs mixed-precision PDIP (BASELINE configs[4] shape by default), HIP-event timed.
python -m tools.ab_mixed [N] [M] [n_obs] [style]"""
import sys

import numpy as np
import torch

from lsc_dr_planner_amd import api, synth


def up(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_obs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    style = sys.argv[4] if len(sys.argv) > 4 else "forest"
    dim, dev = 3, torch.device("cuda", 0)
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=2000 + N, style=style)
    mk = lambda **kw: api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, **kw))  # noqa: E731
    s64 = mk()
    for _ in range(3):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        r = s64.solve_host(hdr, rows, off, sfc, want_info=False, x_init=api.x_init_from_swarm(b, dim))
        assert (r["status"] == 0).all()
        sw.advance(r["x"])
    b = sw.build()
    hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
    x0 = api.x_init_from_swarm(b, dim)
    dh, do_, ds, dxi = up(hdr, dev), up(off, dev), up(sfc, dev), up(x0, dev)
    dx = torch.zeros(N * s64.nv, dtype=torch.float64, device=dev)
    dob = torch.zeros(N, dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    ref = None
    for name, kw in (("fp64", {}), ("fp64 rows_f32", dict(row_format=api.ROWS_F32)), ("mixed", dict(precision=api.PRECISION_MIXED)),
                     ("mixed rows_f32", dict(precision=api.PRECISION_MIXED, row_format=api.ROWS_F32))):
        sol = mk(**kw)
        dr = up(sol.rows_in_format(rows), dev)
        for _ in range(3):
            sol.solve_device(N, sw.n_obs, dh, dr, do_, ds, dx, dob, dst, dinfo, d_x_init=dxi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            sol.solve_device(N, sw.n_obs, dh, dr, do_, ds, dx, dob, dst, dinfo, d_x_init=dxi)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        info = dinfo.cpu().numpy().view(api.INFO_DTYPE)
        x = dx.cpu().numpy().reshape(N, -1)
        if ref is None:
            ref = x.copy()
        print("%-16s %8.4f ms  %.3e QP/s  iters mean %.2f max %d  non-optimal %d  repaired %d  floor %d  max|dx vs fp64| %.2e" % (
            name, ms, N / ms * 1e3, info["iterations"].mean(), info["iterations"].max(), int((dst.cpu().numpy() != 0).sum()),
            int(((info["flags"] & 2) != 0).sum()), int(((info["flags"] & 1) != 0).sum()), np.abs(x - ref).max()))


if __name__ == "__main__":
    main()
