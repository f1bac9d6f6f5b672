"""ctypes binding of the CPU oracle (oracle/lscqp_oracle.c, oracle/lscgen_oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py may import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblscqp_oracle.so")


class OrcClass(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("n", C.c_int), ("phi", C.c_int), ("phi_n", C.c_int), ("dim", C.c_int),
        ("planner_lsc", C.c_int), ("use_sfc", C.c_int),
        ("dt", C.c_double), ("w_c", C.c_double), ("w_t", C.c_double), ("comm_range", C.c_double),
        ("world_min", C.c_double * 3), ("world_max", C.c_double * 3),
    ]


class OrcAgent(C.Structure):
    _fields_ = [
        ("p0", C.c_double * 3), ("v0", C.c_double * 3), ("a0", C.c_double * 3), ("goal", C.c_double * 3),
        ("next_waypoint", C.c_double * 3), ("vmax", C.c_double * 3), ("amax", C.c_double * 3),
        ("radius", C.c_double), ("nominal_velocity", C.c_double), ("n_obs", C.c_int),
    ]


class OrcSizes(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("nv", "neq", "nineq", "n_sfc", "n_lsc", "n_vel", "n_acc", "n_comm", "n_lsc_skipped")]


AGENT_DTYPE = np.dtype([
    ("p0", "f8", 3), ("v0", "f8", 3), ("a0", "f8", 3), ("goal", "f8", 3), ("next_waypoint", "f8", 3),
    ("vmax", "f8", 3), ("amax", "f8", 3), ("radius", "f8"), ("nominal_velocity", "f8"), ("n_obs", "i4"),
], align=True)
LSC_DTYPE = np.dtype([("p", "f8", 3), ("nrm", "f8", 3), ("d", "f8")], align=True)   # 56 B, reference LSC
BOX_DTYPE = np.dtype([("bmin", "f8", 3), ("bmax", "f8", 3)], align=True)

assert AGENT_DTYPE.itemsize == C.sizeof(OrcAgent)

_lib = None


_REF_GJK_PATH = os.path.join(_HERE, "_ref", "libref_gjk.so")
_REFERENCE = "/root/reference"


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("lscqp_oracle.c", "lscgen_oracle.c", "lscgoal_oracle.c", "lscpost_oracle.c", "lscmode_oracle.c", "lscsfc_oracle.c", "lscpred_oracle.c", "Makefile", "lscqp_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liblscqp_oracle.so"])
    return _LIB_PATH


def build_ref():
    """oracle/_ref/libref_gjk.so: the reference's own openGJK compiled from the reference checkout (build container
    only; the GPU box uses the prebuilt file that travelled with the snapshot, or the committed golden vectors).
    Returns the path, or None when neither the checkout nor a prebuilt library is there."""
    if os.path.exists(os.path.join(_REFERENCE, "src", "openGJK", "openGJK.cpp")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return _REF_GJK_PATH if os.path.exists(_REF_GJK_PATH) else None


_ref_gjk = None


def ref_gjk(hull, point=(0.0, 0.0, 0.0)):
    """The reference's openGJK on (hull (k,3), point): returns (distance, closest point of the hull relative to
    `point`), exactly what closestPointsBetweenPointAndConvexHull gets (reference include/geometry.hpp:266-296).
    None if oracle/_ref is not built."""
    global _ref_gjk
    if _ref_gjk is None:
        if not os.path.exists(_REF_GJK_PATH):
            return None
        _ref_gjk = C.CDLL(_REF_GJK_PATH)
        dp = C.POINTER(C.c_double)
        _ref_gjk.ref_gjk_point_hull.restype = C.c_double
        _ref_gjk.ref_gjk_point_hull.argtypes = [dp, C.c_int, dp, dp]
    h = np.ascontiguousarray(hull, dtype=np.float64)
    pt = np.ascontiguousarray(point, dtype=np.float64)
    v = np.zeros(3)
    d = _ref_gjk.ref_gjk_point_hull(_dp(h), h.shape[0], _dp(pt), _dp(v))
    return d, v


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        _lib.orc_solve.restype = C.c_int
        _lib.orc_solve.argtypes = [C.POINTER(OrcClass), C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                   dp, dp, dp, dp, dp, dp, ip]
        _lib.orc_count.argtypes = [C.POINTER(OrcClass), C.c_void_p, C.c_void_p, C.POINTER(OrcSizes)]
        _lib.orc_assemble.argtypes = [C.POINTER(OrcClass), C.c_void_p, C.c_void_p, C.c_void_p] + [dp] * 9
        _lib.orc_kkt.argtypes = [C.POINTER(OrcClass), C.c_void_p, C.c_void_p, C.c_void_p, dp, dp, dp, dp, dp, dp]
        _lib.orc_state_at.argtypes = [C.POINTER(OrcClass), dp, C.c_double, dp, dp, dp]
        _lib.orc_terminal_segments.restype = C.c_int
        _lib.orc_terminal_segments.argtypes = [C.POINTER(OrcClass), C.c_void_p]
        _lib.orc_q_base.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp]
        _lib.orc_aeq_base.restype = C.c_int
        _lib.orc_aeq_base.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp]
        _lib.orc_bernstein.argtypes = [C.c_int, dp]
        _lib.orc_hull_closest_point.restype = C.c_double
        _lib.orc_hull_closest_point.argtypes = [dp, C.c_int, dp]
        _lib.orc_generate_lsc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_void_p]
        _lib.orc_generate_mode.argtypes = [C.c_int] * 6 + [dp, ip, dp, dp, dp, C.c_void_p]
        _lib.orc_sub_segment.argtypes = [dp, C.c_double, C.c_double, dp]
        _lib.orc_shift_prev_plan.argtypes = [C.c_int, C.c_double, dp, dp]
        _lib.orc_const_vel_traj.argtypes = [C.c_int, C.c_double, dp, dp, dp]
        _lib.orc_obstacle_sizes.argtypes = [C.c_int, C.c_void_p, C.c_void_p, dp, C.c_double, dp]
        _lib.orc_generate_lsc_obstacle.argtypes = [C.c_int, C.c_int, C.c_void_p, dp, dp, C.c_double, dp, C.c_double, C.c_void_p, C.c_void_p]
        _lib.orc_select_neighbours.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, dp, ip, ip]
        _lib.orc_segseg_closest.restype = C.c_double
        _lib.orc_segseg_closest.argtypes = [dp] * 6
        _lib.orc_goal_rows.restype = C.c_int
        _lib.orc_goal_rows.argtypes = [C.POINTER(OrcClass), dp, dp, C.c_int, C.c_void_p, C.c_void_p, dp, dp]
        _lib.orc_goal_opt.restype = C.c_int
        _lib.orc_goal_opt.argtypes = [C.POINTER(OrcClass), dp, dp, C.c_int, C.c_void_p, C.c_void_p, dp, C.POINTER(C.c_double)]
        _lib.orc_validate_step.restype = C.c_int
        _lib.orc_validate_step.argtypes = [C.POINTER(OrcClass), C.c_void_p, C.c_void_p, dp, C.c_double, C.c_double, dp]
        _lib.orc_safety_metrics.argtypes = [C.POINTER(OrcClass), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp,
                                            C.c_void_p, dp]
        _lib.orc_safety_obstacles.argtypes = [C.POINTER(OrcClass), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, C.c_int, dp, ip, dp]
        _lib.orc_map_create.restype = C.c_void_p
        _lib.orc_map_create.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double]
        _lib.orc_map_destroy.argtypes = [C.c_void_p]
        _lib.orc_map_info.argtypes = [C.c_void_p, ip, ip]
        _lib.orc_map_occ.restype = C.POINTER(C.c_ubyte)
        _lib.orc_map_occ.argtypes = [C.c_void_p]
        _lib.orc_map_nearest.restype = ip
        _lib.orc_map_nearest.argtypes = [C.c_void_p]
        _lib.orc_construct_sfc_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, dp, C.c_void_p, ip]
        _lib.orc_solve_batch.restype = C.c_int
        _lib.orc_solve_batch.argtypes = [C.POINTER(OrcClass), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_double, C.c_int, C.c_int, dp, dp, ip, ip]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_class(M=5, dim=3, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0, planner_lsc=True, use_sfc=True,
               world_min=(-5, -5, 0), world_max=(5, 5, 2.5), n=5, phi=3, phi_n=1):
    c = OrcClass()
    c.M, c.n, c.phi, c.phi_n, c.dim = M, n, phi, phi_n, dim
    c.planner_lsc, c.use_sfc = int(planner_lsc), int(use_sfc)  # (planner_lsc: True / 1 = LSC, False / 0 = DLSC or BVC, 2 = RECIPROCALRSFC)
    c.dt, c.w_c, c.w_t, c.comm_range = dt, w_c, w_t, comm_range
    for k in range(3):
        c.world_min[k] = world_min[k]
        c.world_max[k] = world_max[k]
    return c


def make_agent(p0, goal, v0=(0, 0, 0), a0=(0, 0, 0), next_waypoint=None, vmax=(1, 1, 1), amax=(2, 2, 2),
               radius=0.15, nominal_velocity=1.0, n_obs=0):
    a = np.zeros((), AGENT_DTYPE)
    a["p0"], a["v0"], a["a0"], a["goal"] = p0, v0, a0, goal
    a["next_waypoint"] = goal if next_waypoint is None else next_waypoint
    a["vmax"], a["amax"] = vmax, amax
    a["radius"], a["nominal_velocity"], a["n_obs"] = radius, nominal_velocity, n_obs
    return a


def count(cls, agent, lsc=None):
    s = OrcSizes()
    lib().orc_count(C.byref(cls), _vp(agent), _vp(lsc), C.byref(s))
    return s


def assemble(cls, agent, lsc=None, sfc=None):
    s = count(cls, agent, lsc)
    nv, neq, mi = s.nv, s.neq, s.nineq
    P = np.zeros((nv, nv)); q = np.zeros(nv); r = np.zeros(1)
    Aeq = np.zeros((max(neq, 1), nv)); beq = np.zeros(max(neq, 1))
    G = np.zeros((max(mi, 1), nv)); h = np.zeros(max(mi, 1))
    lb = np.zeros(nv); ub = np.zeros(nv)
    lib().orc_assemble(C.byref(cls), _vp(agent), _vp(lsc), _vp(sfc), _dp(P), _dp(q), _dp(r), _dp(Aeq), _dp(beq),
                       _dp(G), _dp(h), _dp(lb), _dp(ub))
    return dict(P=P, q=q, r=float(r[0]), Aeq=Aeq[:neq], beq=beq[:neq], G=G[:mi], h=h[:mi], lb=lb, ub=ub, sizes=s)


def solve(cls, agent, lsc=None, sfc=None, tol=1e-11, max_iter=200):
    s = count(cls, agent, lsc)
    x = np.zeros(s.nv); obj = np.zeros(1)
    y = np.zeros(max(s.neq, 1)); lam = np.zeros(max(s.nineq, 1)); mlb = np.zeros(s.nv); mub = np.zeros(s.nv)
    it = C.c_int(0)
    st = lib().orc_solve(C.byref(cls), _vp(agent), _vp(lsc), _vp(sfc), tol, max_iter, _dp(x), _dp(obj), _dp(y),
                         _dp(lam), _dp(mlb), _dp(mub), C.byref(it))
    return dict(status=st, x=x, obj=float(obj[0]), y=y[:s.neq], lam=lam[:s.nineq], mu_lb=mlb, mu_ub=mub,
                iters=it.value)


def kkt(cls, agent, lsc, sfc, x, y=None, lam=None, mu_lb=None, mu_ub=None):
    res = np.zeros(6)
    null = C.POINTER(C.c_double)()
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_kkt(C.byref(cls), _vp(agent), _vp(lsc), _vp(sfc), _dp(x),
                  null if y is None else _dp(y), null if lam is None else _dp(lam),
                  null if mu_lb is None else _dp(mu_lb), null if mu_ub is None else _dp(mu_ub), _dp(res))
    return dict(stationarity=res[0], eq=res[1], ineq=res[2], neg_mult=res[3], comp=res[4], obj=res[5])


def state_at(cls, x, t):
    p = np.zeros(3); v = np.zeros(3); a = np.zeros(3)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_state_at(C.byref(cls), _dp(x), float(t), _dp(p), _dp(v), _dp(a))
    return p[:cls.dim], v[:cls.dim], a[:cls.dim]


def terminal_segments(cls, agent):
    return lib().orc_terminal_segments(C.byref(cls), _vp(agent))


def q_base(n=5, phi=3, phi_n=1, dt=0.2):
    Q = np.zeros((n + 1, n + 1))
    lib().orc_q_base(n, phi, phi_n, dt, _dp(Q))
    return Q


def aeq_base(M, n=5, phi=3, dt=0.2):
    A = np.zeros((max((M - 2) * phi, 1), M * (n + 1)))
    rc = lib().orc_aeq_base(M, n, phi, dt, _dp(A))
    return rc, A[:(M - 2) * phi]


def bernstein(n=5):
    B = np.zeros((n + 1, n + 1))
    lib().orc_bernstein(n, _dp(B))
    return B


def solve_batch(cls, agents, lsc=None, lsc_off=None, sfc=None, tol=1e-11, max_iter=200, threads=1):
    """agents: AGENT_DTYPE[n]; lsc: LSC_DTYPE[total]; lsc_off: int64[n] start index per agent; sfc: BOX_DTYPE[n*M]."""
    n = len(agents)
    nv = cls.dim * cls.M * (cls.n + 1)
    x = np.zeros((n, nv)); obj = np.zeros(n)
    status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32)
    agents = np.ascontiguousarray(agents)
    if lsc_off is not None:
        lsc_off = np.ascontiguousarray(lsc_off, dtype=np.int64)
    bad = lib().orc_solve_batch(C.byref(cls), n, _vp(agents), _vp(lsc), _vp(lsc_off), _vp(sfc), tol, max_iter,
                                threads, _dp(x), _dp(obj), status.ctypes.data_as(C.POINTER(C.c_int)),
                                iters.ctypes.data_as(C.POINTER(C.c_int)))
    return dict(x=x, obj=obj, status=status, iters=iters, bad=bad)


def hull_closest_point(pts):
    """Closest point to the origin on conv(pts), pts (k,3), k <= 8: (distance, point).  oracle/lscgen_oracle.c."""
    p = np.ascontiguousarray(pts, dtype=np.float64)
    out = np.zeros(3)
    d = lib().orc_hull_closest_point(_dp(p), p.shape[0], _dp(out))
    return d, out


def generate_lsc(traj, neighbours, radius, downwash, goal, dim=3, first_agent=0):
    """Restatement of TrajPlanner::generateLSC for agent obstacles (reference src/traj_planner.cpp:611-657).
    traj (n_total, M, 6, 3) initial / predicted control points, neighbours (n_agents, n_obs) global ids (< 0: none),
    radius, downwash (n_total,), goal (n_agents, 3).  Returns LSC_DTYPE[n_agents, n_obs, M, 6]."""
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    nb = np.ascontiguousarray(neighbours, dtype=np.int32)
    n_total, M = traj.shape[0], traj.shape[1]
    n_agents, n_obs = nb.shape
    r = np.ascontiguousarray(np.broadcast_to(radius, (n_total,)), dtype=np.float64)
    dw = np.ascontiguousarray(np.broadcast_to(downwash, (n_total,)), dtype=np.float64)
    g = np.ascontiguousarray(goal, dtype=np.float64).reshape(n_agents, 3)
    out = np.zeros((n_agents, n_obs, M, 6), LSC_DTYPE)
    lib().orc_generate_lsc(M, dim, n_agents, n_obs, first_agent, _dp(traj), nb.ctypes.data_as(C.POINTER(C.c_int)), _dp(r),
                           _dp(dw), _dp(g), out.ctypes.data_as(C.c_void_p))
    return out


MODE_LSC, MODE_CLSC, MODE_BVC = 0, 1, 2


OBSTACLE_DTYPE = np.dtype([("position", "f8", 3), ("velocity", "f8", 3), ("radius", "f8"), ("downwash", "f8"), ("max_acc", "f8"),
                           ("type", "i4"), ("pad", "i4")])  # orc_obstacle == lscqp_obstacle


class ObsParam(C.Structure):  # orc_obs_param
    _fields_ = [("dt", C.c_double), ("obs_uncertainty_horizon", C.c_double), ("velocity_guard_ratio", C.c_double),
                ("obs_downwash_threshold", C.c_double), ("reset_threshold", C.c_double), ("obs_size_prediction", C.c_int),
                ("use_velocity_guard", C.c_int)]


def obs_param(dt=0.2, obs_uncertainty_horizon=1.0, velocity_guard_ratio=0.75, obs_downwash_threshold=3.0, reset_threshold=0.1,
              obs_size_prediction=True, use_velocity_guard=True):
    """Defaults of src/param.cpp:63-67, 106-108."""
    return ObsParam(dt, obs_uncertainty_horizon, velocity_guard_ratio, obs_downwash_threshold, reset_threshold,
                    int(obs_size_prediction), int(use_velocity_guard))


def sub_segment(cp, t0, tf):
    """Segment<point3d>::subSegment (reference src/trajectory.cpp:15-49): cp (6, 3) -> (6, 3)."""
    cp = np.ascontiguousarray(cp, dtype=np.float64).reshape(6, 3)
    out = np.zeros((6, 3))
    lib().orc_sub_segment(_dp(cp), C.c_double(t0), C.c_double(tf), _dp(out))
    return out


def shift_prev_plan(prev, fraction):
    """initialTrajPlanningPrevSol (reference src/traj_planner.cpp:399-423): prev (N, M, 6, 3) -> (N, M, 6, 3); fraction =
    multisim_time_step / dt (1: shift by a segment, < 1: sub-segment of segment 0)."""
    prev = np.ascontiguousarray(prev, dtype=np.float64)
    out = np.zeros_like(prev)
    for a in range(prev.shape[0]):
        pa = np.ascontiguousarray(prev[a])
        oa = np.zeros_like(pa)
        lib().orc_shift_prev_plan(prev.shape[1], C.c_double(fraction), _dp(pa), _dp(oa))
        out[a] = oa
    return out


def const_vel_traj(M, dt, pos, vel):
    out = np.zeros((M, 6, 3))
    lib().orc_const_vel_traj(M, C.c_double(dt), _dp(np.ascontiguousarray(pos, dtype=np.float64)), _dp(np.ascontiguousarray(vel, dtype=np.float64)), _dp(out))
    return out


def obstacle_sizes(M, param, obstacle, v_agent, amax0):
    out = np.zeros((M, 6))
    o = np.ascontiguousarray(obstacle, dtype=OBSTACLE_DTYPE).reshape(1)
    lib().orc_obstacle_sizes(M, C.byref(param), o.ctypes.data_as(C.c_void_p), _dp(np.ascontiguousarray(v_agent, dtype=np.float64)),
                             C.c_double(amax0), _dp(out))
    return out


def generate_lsc_obstacles(param, own_traj, goal, r_own, v_agent, amax0, obstacles, dim=3):
    """generateLSC for non-agent obstacles (reference src/traj_planner.cpp:611-657 with the prediction / size steps before it):
    own_traj (M, 6, 3), obstacles OBSTACLE_DTYPE[n] -> LSC_DTYPE[n, M, 6]."""
    own = np.ascontiguousarray(own_traj, dtype=np.float64)
    M = own.shape[0]
    obs = np.ascontiguousarray(obstacles, dtype=OBSTACLE_DTYPE).reshape(-1)
    out = np.zeros((len(obs), M, 6), LSC_DTYPE)
    for i in range(len(obs)):
        oi = np.zeros((M, 6), LSC_DTYPE)
        lib().orc_generate_lsc_obstacle(M, dim, C.byref(param), _dp(own), _dp(np.ascontiguousarray(goal, dtype=np.float64)), C.c_double(r_own),
                                        _dp(np.ascontiguousarray(v_agent, dtype=np.float64)), C.c_double(amax0),
                                        obs[i:i + 1].ctypes.data_as(C.c_void_p), oi.ctypes.data_as(C.c_void_p))
        out[i] = oi
    return out


def generate_constraints(mode, traj, neighbours, radius, downwash, goal_all, dim=3, first_agent=0):
    """Restatement of the planner's linear-constraint generators for agent obstacles: mode 0 = generateLSC
    (reference src/traj_planner.cpp:611-657), 1 = generateCLSC (:659-706, the default launch), 2 = generateBVC (:708-734).
    goal_all (n_total, 3): every agent's current goal point.  Returns LSC_DTYPE[n_agents, n_obs, M, 6]."""
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    nb = np.ascontiguousarray(neighbours, dtype=np.int32)
    n_total, M = traj.shape[0], traj.shape[1]
    n_agents, n_obs = nb.shape
    g = np.ascontiguousarray(goal_all, dtype=np.float64).reshape(n_total, 3)
    if mode == MODE_LSC:
        return generate_lsc(traj, nb, radius, downwash, g[first_agent:first_agent + n_agents], dim=dim, first_agent=first_agent)
    r = np.ascontiguousarray(np.broadcast_to(radius, (n_total,)), dtype=np.float64)
    dw = np.ascontiguousarray(np.broadcast_to(downwash, (n_total,)), dtype=np.float64)
    out = np.zeros((n_agents, n_obs, M, 6), LSC_DTYPE)
    lib().orc_generate_mode(mode, M, dim, n_agents, n_obs, first_agent, _dp(traj), nb.ctypes.data_as(C.POINTER(C.c_int)), _dp(r),
                            _dp(dw), _dp(g), out.ctypes.data_as(C.c_void_p))
    return out


def segseg_closest(l1s, l1e, l2s, l2e):
    """closestPointsBetweenLineSegments (reference include/geometry.hpp:174-263) in its float32 arithmetic:
    returns (dist, closest_point1 on line1, closest_point2 on line2)."""
    a = [np.ascontiguousarray(np.float32(v), dtype=np.float64) for v in (l1s, l1e, l2s, l2e)]
    c1, c2 = np.zeros(3), np.zeros(3)
    d = lib().orc_segseg_closest(_dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(c1), _dp(c2))
    return d, c1, c2


def goal_rows(cls, goal, next_waypoint, lsc=None, sfc_last=None):
    """Rows a t + c >= 0 of GoalOptimizer::populatebyrow (reference src/goal_optimizer.cpp:118-155), in its order."""
    g = np.ascontiguousarray(goal, dtype=np.float64)
    w = np.ascontiguousarray(next_waypoint, dtype=np.float64)
    n_obs = 0 if lsc is None else lsc.shape[0]
    lscc = None if lsc is None else np.ascontiguousarray(lsc, dtype=LSC_DTYPE)
    box = None if sfc_last is None else np.ascontiguousarray(sfc_last, dtype=BOX_DTYPE).reshape(1)
    a = np.zeros(2 * cls.dim + n_obs + 2)
    c = np.zeros_like(a)
    nr = lib().orc_goal_rows(C.byref(cls), _dp(g), _dp(w), n_obs, _vp(lscc), _vp(box), _dp(a), _dp(c))
    return a[:nr], c[:nr]


def goal_opt(cls, goal, next_waypoint, lsc=None, sfc_last=None):
    """GoalOptimizer::solve restated: returns (status 0/1, optimised goal (3,), t)."""
    g = np.ascontiguousarray(goal, dtype=np.float64)
    w = np.ascontiguousarray(next_waypoint, dtype=np.float64)
    n_obs = 0 if lsc is None else lsc.shape[0]
    lscc = None if lsc is None else np.ascontiguousarray(lsc, dtype=LSC_DTYPE)
    box = None if sfc_last is None else np.ascontiguousarray(sfc_last, dtype=BOX_DTYPE).reshape(1)
    out = np.zeros(3)
    t = C.c_double(0)
    st = lib().orc_goal_opt(C.byref(cls), _dp(g), _dp(w), n_obs, _vp(lscc), _vp(box), _dp(out), C.byref(t))
    return st, out, t.value


def validate_step(cls, agent, sfc, x, time_step, z_2d=1.0):
    """isSolValid + doStep restated (reference src/traj_planner.cpp:990-1045, src/agent_manager.cpp:29-50):
    returns (valid 0/1, state (9,) = position, velocity, acceleration at time_step)."""
    ag = np.ascontiguousarray(agent, dtype=AGENT_DTYPE).reshape(1)
    box = None if sfc is None else np.ascontiguousarray(sfc, dtype=BOX_DTYPE).reshape(-1)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    st = np.zeros(9)
    ok = lib().orc_validate_step(C.byref(cls), _vp(ag), _vp(box), _dp(xx), float(time_step), float(z_2d), _dp(st))
    return ok, st


def safety_metrics(cls, agents, x_all, radius, downwash, n_samples, step, first=0, z_2d=1.0):
    """MultiSyncSimulator::update's safety ratio / excess ratios (reference src/multi_sync_simulator.cpp:486-577) for the
    local agents `agents` (AGENT_DTYPE, ids first..) among x_all (n_total, nv).  Returns (n_agents, 9):
    ratio, closest j, sample, vel_excess[3], acc_excess[3]."""
    ag = np.ascontiguousarray(agents, dtype=AGENT_DTYPE)
    xx = np.ascontiguousarray(x_all, dtype=np.float64)
    n_total = xx.shape[0]
    r = np.ascontiguousarray(np.broadcast_to(radius, (n_total,)), dtype=np.float64)
    dw = np.ascontiguousarray(np.broadcast_to(downwash, (n_total,)), dtype=np.float64)
    out = np.zeros((ag.shape[0], 9))
    lib().orc_safety_metrics(C.byref(cls), ag.shape[0], first, n_total, int(n_samples), float(step), float(z_2d), _dp(xx), _dp(r), _dp(dw),
                             _vp(ag), _dp(out))
    return out


def safety_obstacles(cls, n_agents, x_all, radius, downwash, obstacles, n_samples, step, first=0, z_2d=1.0, skip=None):
    """The obstacle leg of MultiSyncSimulator::update's safety figures (reference src/multi_sync_simulator.cpp:527-557): obstacles
    (n_obs, 5) = x, y, z, radius, downwash; skip (n_obs,) != 0 marks "real" obstacles.  Returns (n_agents, 3): ratio, obstacle, sample."""
    xx = np.ascontiguousarray(x_all, dtype=np.float64)
    n_total = xx.shape[0]
    r = np.ascontiguousarray(np.broadcast_to(radius, (n_total,)), dtype=np.float64)
    dw = np.ascontiguousarray(np.broadcast_to(downwash, (n_total,)), dtype=np.float64)
    ob = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 5)
    sk = np.ascontiguousarray(np.zeros(len(ob)) if skip is None else skip, dtype=np.int32)
    out = np.zeros((int(n_agents), 3))
    lib().orc_safety_obstacles(C.byref(cls), int(n_agents), int(first), int(n_samples), float(step), float(z_2d), _dp(xx), _dp(r), _dp(dw),
                               len(ob), _dp(ob), sk.ctypes.data_as(C.POINTER(C.c_int)), _dp(out))
    return out


SFC_INIT, SFC_FROM_HULL, SFC_FROM_POINT = 0, 1, 2


class Map:
    """The occupancy grid + nearest-occupied-cell field of the corridor construction (oracle/lscsfc_oracle.c):
    boxes (n, 6) = centre xyz, size xyz, the rows of the reference's world CSV (src/map_manager.cpp:262-305)."""

    def __init__(self, boxes, world_min, world_max, res=0.1, max_dist=1.0):
        b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        wmin = np.ascontiguousarray(world_min, dtype=np.float64)
        wmax = np.ascontiguousarray(world_max, dtype=np.float64)
        self._h = lib().orc_map_create(_dp(b), b.shape[0], _dp(wmin), _dp(wmax), float(res), float(max_dist))
        dims, key0 = np.zeros(3, np.int32), np.zeros(3, np.int32)
        lib().orc_map_info(self._h, dims.ctypes.data_as(C.POINTER(C.c_int)), key0.ctypes.data_as(C.POINTER(C.c_int)))
        self.dims, self.key0, self.res = dims, key0, res

    def occ(self):
        n = int(np.prod(self.dims))
        return np.ctypeslib.as_array(lib().orc_map_occ(self._h), shape=(n,)).reshape(self.dims[2], self.dims[1], self.dims[0]).copy()

    def nearest(self):
        n = int(np.prod(self.dims))
        return np.ctypeslib.as_array(lib().orc_map_nearest(self._h), shape=(n,)).reshape(self.dims[2], self.dims[1], self.dims[0]).copy()

    def construct_sfc(self, mode, pts, radius, sfc):
        """Corridor update of n agents: pts (n, 3, 3) = (position | last point, goal point, next waypoint), sfc BOX_DTYPE
        (n, M) updated in place (mode SFC_INIT fills it).  Returns status (n,): 1 = new box, 0 = previous kept / failure."""
        P = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 9)
        n = P.shape[0]
        assert sfc.dtype == BOX_DTYPE and sfc.flags.c_contiguous and sfc.shape[0] == n
        r = np.ascontiguousarray(np.broadcast_to(radius, (n,)), dtype=np.float64)
        st = np.zeros(n, np.int32)
        lib().orc_construct_sfc_batch(self._h, int(mode), sfc.shape[1], n, _dp(P), _dp(r), sfc.ctypes.data_as(C.c_void_p),
                                      st.ctypes.data_as(C.POINTER(C.c_int)))
        return st

    def __del__(self):
        try:
            lib().orc_map_destroy(self._h)
        except Exception:
            pass


def select_neighbours(pos, n_obs, comm_range, first=0, n_agents=None):
    """broadcastMsgs' range filter (reference src/multi_sync_simulator.cpp:318-333): (nbr (n_agents, n_obs) int32 with -1
    padding, count (n_agents,)).  More than n_obs in range: the n_obs nearest are kept (the build's capacity rule)."""
    P = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    n_total = P.shape[0]
    n_agents = n_total - first if n_agents is None else n_agents
    nbr = np.zeros((n_agents, n_obs), np.int32)
    cnt = np.zeros(n_agents, np.int32)
    ip = C.POINTER(C.c_int)
    lib().orc_select_neighbours(n_agents, first, n_total, n_obs, float(comm_range), _dp(P), nbr.ctypes.data_as(ip), cnt.ctypes.data_as(ip))
    return nbr, cnt
