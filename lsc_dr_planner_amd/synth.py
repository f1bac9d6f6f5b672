"""Synthetic swarms that produce trajectory-QP batches of the shapes BASELINE.json names.

This is workload generation for tests and bench.py, not part of the solver.  It restates just
enough of the reference's per-replan pipeline to make the QPs look like the reference's:

  * initial trajectory  = previous solution shifted by one segment      (src/traj_planner.cpp:399-411)
  * obstacle prediction = neighbours' previous solutions, shifted        (src/traj_planner.cpp:273-310)
  * LSC rows            = generateLSC                                    (src/traj_planner.cpp:611-657):
        downwash-scaled coordinates, normal = unit vector from the origin to the closest point of the convex
        hull of the 6 relative control points (normalVectorBetweenPolys, :1179-1205),
        d_i = 1/2 (r_i + r_j + rel_i . normal), then normal.z /= downwash
  * SFC                 = one axis-aligned box per segment, snapped to the 0.1 m grid like expandSFC
                          (src/collision_constraints.cpp:820-881), always containing the initial trajectory
  * goal / next waypoint = one 0.5 m grid step towards the final goal (launch/simulation.launch:88)

All values are rounded to float32 and widened, mimicking octomap::point3d.
The solver that carries the swarm forward is supplied by the caller (the CPU oracle in tests, the HIP solver in
bench.py); the generator itself never solves anything.
"""
import itertools

import numpy as np

LSC_DTYPE = np.dtype([("p", "f8", 3), ("nrm", "f8", 3), ("d", "f8")], align=True)
BOX_DTYPE = np.dtype([("bmin", "f8", 3), ("bmax", "f8", 3)], align=True)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _closest_point_hull_origin(pts):
    """pts: (..., K, 3). Closest point to the origin on conv(pts), brute force over vertex/edge/triangle faces.
    Returns (closest (...,3), dist (...))."""
    K = pts.shape[-2]
    lead = pts.shape[:-2]
    best_d = np.full(lead, np.inf)
    best_p = np.zeros(lead + (3,))

    def consider(p, valid):
        d = np.linalg.norm(p, axis=-1)
        upd = valid & (d < best_d)
        best_d[upd] = d[upd]
        best_p[upd] = p[upd]

    for i in range(K):
        consider(pts[..., i, :], np.ones(lead, bool))
    for i, j in itertools.combinations(range(K), 2):
        a, b = pts[..., i, :], pts[..., j, :]
        ab = b - a
        den = np.einsum("...k,...k->...", ab, ab)
        ok = den > 1e-18
        t = np.where(ok, -np.einsum("...k,...k->...", a, ab) / np.where(ok, den, 1.0), 0.0)
        consider(a + t[..., None] * ab, ok & (t >= 0) & (t <= 1))
    for i, j, k in itertools.combinations(range(K), 3):
        a, b, c = pts[..., i, :], pts[..., j, :], pts[..., k, :]
        e1, e2 = b - a, c - a
        g11 = np.einsum("...k,...k->...", e1, e1)
        g12 = np.einsum("...k,...k->...", e1, e2)
        g22 = np.einsum("...k,...k->...", e2, e2)
        r1 = -np.einsum("...k,...k->...", a, e1)
        r2 = -np.einsum("...k,...k->...", a, e2)
        det = g11 * g22 - g12 * g12
        ok = det > 1e-14 * np.maximum(g11 * g22, 1e-300)
        dets = np.where(ok, det, 1.0)
        u = (r1 * g22 - r2 * g12) / dets
        v = (r2 * g11 - r1 * g12) / dets
        consider(a + u[..., None] * e1 + v[..., None] * e2, ok & (u >= 0) & (v >= 0) & (u + v <= 1))
    return best_p, best_d


class Swarm:
    """N agents, synchronous replanning every dt, as in MultiSyncSimulator::run (src/multi_sync_simulator.cpp:81-129)."""

    def __init__(self, N, M=5, dim=3, n_obs=20, seed=0, dt=0.2, radius=0.15, downwash=2.0, vmax=1.0, amax=2.0,
                 nominal_velocity=1.0, comm_range=3.0, style="forest", spacing=None, world_margin=2.0, neighbour_order="distance"):
        self.N, self.M, self.dim, self.dt, self.n = N, M, dim, dt, 5
        self.n_obs = min(n_obs, N - 1)
        self.radius, self.downwash = radius, downwash
        self.vmax, self.amax, self.nominal_velocity = vmax, amax, nominal_velocity
        self.comm_range = comm_range
        self.style = style
        # "distance": nearest first (rounds 1-5's batches).  "id": the same n_obs nearest agents, listed by agent id -- the order the
        # reference's obstacle list has (broadcastMsgs walks the agents in mission order, src/multi_sync_simulator.cpp:305-352), stable from
        # replan to replan, which is what a record of active rows carried across replans relies on (lscqp_solve_batch_device_hinted)
        self.neighbour_order = neighbour_order
        rng = np.random.default_rng(seed)
        self.rng = rng
        # jittered lattice: min separation comfortably above 2.2 r in downwash-scaled coordinates
        if spacing is None:
            spacing = 0.9 if style == "forest" else 0.75
        zsp = spacing * downwash if dim == 3 else 0.0
        if dim == 3:
            g = int(np.ceil(N ** (1 / 3)))
            idx = np.stack(np.unravel_index(rng.permutation(g ** 3)[:N], (g, g, g)), axis=-1).astype(float)
            base = idx * np.array([spacing, spacing, zsp])
            jit = (rng.random((N, 3)) - 0.5) * np.array([spacing, spacing, zsp]) * 0.35
        else:
            g = int(np.ceil(N ** 0.5))
            idx = np.stack(np.unravel_index(rng.permutation(g * g)[:N], (g, g)), axis=-1).astype(float)
            base = np.concatenate([idx * spacing, np.full((N, 1), 1.0)], axis=1)
            jit = np.concatenate([(rng.random((N, 2)) - 0.5) * spacing * 0.35, np.zeros((N, 1))], axis=1)
        pos = f32(base + jit)
        ext = pos.max(0) - pos.min(0)
        self.world_min = f32(pos.min(0) - world_margin)
        self.world_max = f32(pos.max(0) + world_margin)
        if dim == 2:
            self.world_min[2], self.world_max[2] = 0.0, 2.5
        # final goals: a random permutation of the start positions (antipodal-like exchange)
        self.final_goal = pos[rng.permutation(N)].copy()
        self.pos = pos
        self.vel = np.zeros((N, 3))
        self.acc = np.zeros((N, 3))
        self.prev = None  # (N, M, 6, 3) previous solution control points
        self.step = 0
        self._ext = ext

    # -------- trajectories ------------------------------------------------------------------------------
    def initial_traj(self):
        """(N, M, 6, 3): src/traj_planner.cpp:391-423 (planner_seq < 2 -> constant velocity; else shifted prev)."""
        N, M = self.N, self.M
        if self.prev is None:
            t = (np.arange(M * 6) * (self.dt / 5)).reshape(M, 6)  # planConstVelTraj, src/trajectory.cpp:79-91
            return f32(self.pos[:, None, None, :] + self.vel[:, None, None, :] * t[None, :, :, None])
        it = np.empty_like(self.prev)
        it[:, :-1] = self.prev[:, 1:]
        it[:, -1] = self.prev[:, -1, 5][:, None, :]
        return it

    # -------- constraints -------------------------------------------------------------------------------
    def neighbours(self):
        """n_obs nearest agents by L-infinity distance (the reference filters by L-inf range,
        src/multi_sync_simulator.cpp:319-333)."""
        d = np.abs(self.pos[:, None, :] - self.pos[None, :, :]).max(-1)
        np.fill_diagonal(d, np.inf)
        nb = np.argsort(d, axis=1, kind="stable")[:, : self.n_obs]
        return np.sort(nb, axis=1) if self.neighbour_order == "id" else nb

    def build_lsc(self, init, nbr):
        N, M, K = self.N, self.M, self.n_obs
        dw = self.downwash if self.dim == 3 else 1.0  # equal radii/downwash -> downwashBetween == downwash
        obs = init[nbr]  # (N, K, M, 6, 3) neighbours' predicted control points
        scale = np.array([1.0, 1.0, 1.0 / dw])
        rel = f32((init[:, None] - obs) * scale)  # transformed relative control points
        if self.dim == 2:
            rel[..., 2] = 0.0
        cp, dist = _closest_point_hull_origin(rel)
        nrm = cp / np.maximum(dist, 1e-12)[..., None]
        nrm = f32(nrm)
        dvals = 0.5 * (2 * self.radius + np.einsum("nkmic,nkmc->nkmi", rel, nrm))
        nrm_out = nrm.copy()
        nrm_out[..., 2] = nrm_out[..., 2] / dw
        nrm_out = f32(nrm_out)
        lsc = np.zeros((N, K, M, 6), LSC_DTYPE)
        lsc["p"] = f32(obs)
        lsc["nrm"] = nrm_out[:, :, :, None, :]
        lsc["d"] = dvals
        return lsc, dist

    def build_sfc(self, init):
        N, M = self.N, self.M
        rng = self.rng
        lo = init.min(axis=2)  # (N, M, 3)
        hi = init.max(axis=2)
        if self.style == "maze":
            # corridors: narrow (0.7-1.0 m wide) in one or two axes, long in the others
            grow = rng.uniform(0.8, 3.0, (N, 1, 3, 2)).repeat(M, axis=1)
            narrow = rng.random((N, 1, 3)) < 0.5
            narrow[..., 2] |= ~narrow.any(-1)
            w = rng.uniform(0.35, 0.5, (N, 1, 3, 2))
            grow = np.where(narrow[..., None].repeat(M, axis=1), w.repeat(M, axis=1), grow)
        else:
            grow = rng.uniform(0.5, 2.0, (N, 1, 3, 2)).repeat(M, axis=1)
        bmin = np.floor((lo - grow[..., 0]) * 10) / 10 + 0.05
        bmax = np.ceil((hi + grow[..., 1]) * 10) / 10 - 0.05
        bmin = np.minimum(bmin, lo)
        bmax = np.maximum(bmax, hi)
        bmin = np.maximum(bmin, self.world_min)
        bmax = np.minimum(bmax, self.world_max)
        sfc = np.zeros((N, M), BOX_DTYPE)
        sfc["bmin"] = f32(bmin) - 0.0
        sfc["bmax"] = f32(bmax) + 0.0
        # float32 rounding must not cut into the initial trajectory
        sfc["bmin"] = np.minimum(sfc["bmin"], lo)
        sfc["bmax"] = np.maximum(sfc["bmax"], hi)
        return sfc

    def goals(self, sfc):
        """one 0.5 m L-inf grid step towards the final goal, clipped into the last SFC box."""
        d = self.final_goal - self.pos
        step = np.clip(d, -0.5, 0.5)
        wp = self.pos + step
        wp = np.clip(wp, sfc["bmin"][:, -1], sfc["bmax"][:, -1])
        if self.dim == 2:
            wp[:, 2] = self.pos[:, 2]
        return f32(wp)

    def build(self):
        """Returns dict(header arrays, lsc (N,K,M,6) LSC_DTYPE, sfc (N,M) BOX_DTYPE, init (N,M,6,3))."""
        init = self.initial_traj()
        nbr = self.neighbours()
        lsc, dist = self.build_lsc(init, nbr)
        sfc = self.build_sfc(init)
        wp = self.goals(sfc)
        return dict(p0=self.pos.copy(), v0=self.vel.copy(), a0=self.acc.copy(), goal=wp.copy(), next_waypoint=wp.copy(),
                    lsc=lsc, sfc=sfc, init=init, min_hull_dist=dist.min() if dist.size else np.inf, nbr=nbr)

    # -------- advance -----------------------------------------------------------------------------------
    def advance(self, x):
        """x: (N, dim*M*6) solved control points in the reference variable order.  State := trajectory at t = dt
        (AgentManager::doStep, src/agent_manager.cpp:29-50), rounded to float32 like desired_traj."""
        N, M, dim = self.N, self.M, self.dim
        cps = np.asarray(x).reshape(N, dim, M, 6).transpose(0, 2, 3, 1)
        full = np.empty((N, M, 6, 3))
        full[..., :dim] = cps
        if dim == 2:
            full[..., 2] = self.pos[:, None, None, 2]
        full = f32(full)
        self.prev = full
        s1 = full[:, 1] if M > 1 else full[:, 0]
        self.pos = s1[:, 0].copy()
        self.vel = f32((s1[:, 1] - s1[:, 0]) * (5 / self.dt))
        self.acc = f32((s1[:, 2] - 2 * s1[:, 1] + s1[:, 0]) * (20 / self.dt ** 2))
        self.step += 1
