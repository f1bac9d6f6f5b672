"""ctypes binding of the C ABI (include/lscqp.h -> liblscqp.so).  Plumbing for tests and bench.py only.

The product is the shared library (HIP kernels + C ABI) and the C++ shim in lsc_dr_planner_amd/shim/.  This module
only moves bytes: numpy structured arrays for host calls, torch CUDA tensors (raw device pointers) for resident
calls.  It never computes anything and has NO CPU fallback: if liblscqp.so is missing, importing fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSCQP_LIB") or os.path.join(_HERE, "liblscqp.so")  # (LSCQP_LIB: A/B builds of the development tools)

STATUS_OPTIMAL, STATUS_INFEASIBLE, STATUS_ITER_LIMIT, STATUS_NUMERIC, STATUS_CAPACITY = 0, 1, 2, 3, 4
PRECISION_F64, PRECISION_MIXED = 0, 1  # lscqp_class_desc.precision
WARM_DEFAULT, WARM_TIGHT = 0, 1  # lscqp_class_desc.warm_start
INFO_FLOOR_ACCEPTED, INFO_REPAIRED, INFO_RECENTRED, INFO_REMEMBERED, INFO_SHIFTED, INFO_RESCUED, INFO_ACTIVE_SET = 1, 2, 4, 8, 16, 32, 64  # lscqp_info.flags
ACTIVE_SET_DEFAULT, ACTIVE_SET_OFF, ACTIVE_SET_ONLY = 0, 1, 2  # lscqp_class_desc.active_set
(DAS_WHY_CAPACITY, DAS_WHY_EMPTY_INTERVAL, DAS_WHY_ROWS, DAS_WHY_STEPS, DAS_WHY_NO_STEP, DAS_WHY_PIVOT, DAS_WHY_VERIFICATION,
 DAS_WHY_MULTIPLIER) = range(1, 9)
PLANNER_DLSC, PLANNER_LSC, PLANNER_BVC, PLANNER_RSFC = 0, 1, 2, 3
OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP = 0, 1, 2, 3, 4
SFC_INIT, SFC_FROM_HULL, SFC_FROM_POINT = 0, 1, 2  # lscqp_construct_sfc_device modes
GEN_LSC, GEN_CLSC, GEN_BVC = 0, 1, 2  # lscqp_generate_constraints_device modes (include/lscqp.h)

HEADER_DTYPE = np.dtype([
    ("p0", "f8", 3), ("v0", "f8", 3), ("a0", "f8", 3), ("goal", "f8", 3), ("next_waypoint", "f8", 3),
    ("vmax", "f8", 3), ("amax", "f8", 3), ("radius", "f8"), ("nominal_velocity", "f8"),
    ("n_obs", "i4"), ("terminal_segments", "i4"), ("reserved", "u4", 2), ("pad", "f8", 7),
])
ROW_DTYPE = np.dtype([("nx", "f8"), ("ny", "f8"), ("nz", "f8"), ("b", "f8")])
ROW_F32_DTYPE = np.dtype([("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("b", "f4")])  # lscqp_row_f32 (row_format = ROWS_F32)
ROWS_F64, ROWS_F32 = 0, 1
BOX_DTYPE = np.dtype([("bmin", "f8", 3), ("bmax", "f8", 3)])
SAFETY_DTYPE = np.dtype([("safety_ratio", "f8"), ("closest_agent", "i4"), ("sample", "i4"), ("vel_excess_ratio", "f8", 3),
                         ("acc_excess_ratio", "f8", 3)])
SAFETY_OBS_DTYPE = np.dtype([("safety_ratio_obs", "f8"), ("closest_obstacle", "i4"), ("sample", "i4")])  # lscqp_safety_obs
OBSTACLE_REAL = 2
OBSTACLE_DTYPE = np.dtype([("position", "f8", 3), ("velocity", "f8", 3), ("radius", "f8"), ("downwash", "f8"), ("max_acc", "f8"),
                           ("type", "i4"), ("reserved", "i4")])  # lscqp_obstacle


class ObstacleParam(C.Structure):  # lscqp_obstacle_param
    _fields_ = [("obs_uncertainty_horizon", C.c_double), ("velocity_guard_ratio", C.c_double), ("obs_downwash_threshold", C.c_double),
                ("reset_threshold", C.c_double), ("obs_size_prediction", C.c_int32), ("use_velocity_guard", C.c_int32)]


INFO_DTYPE = np.dtype([("iterations", "i4"), ("flags", "i4"), ("res_primal", "f8"), ("res_dual", "f8"),
                       ("gap", "f8")])
assert HEADER_DTYPE.itemsize == 256 and ROW_DTYPE.itemsize == 32 and BOX_DTYPE.itemsize == 48
assert INFO_DTYPE.itemsize == 32
# lscqp_diag (failure diagnostics): per row family the largest violation / count, and the most violated row by name
ROW_BOUND, ROW_SFC, ROW_LSC, ROW_VEL, ROW_ACC, ROW_COMM_PAIR, ROW_COMM_WAYPOINT, ROW_EQUALITY, ROW_FAMILIES = range(9)
DIAG_DTYPE = np.dtype([("worst", "f8", 8), ("violated", "i4", 8), ("violation", "f8"), ("family", "i4"), ("obstacle", "i4"),
                       ("segment", "i4"), ("point", "i4"), ("axis", "i4"), ("reserved", "i4")])
assert DIAG_DTYPE.itemsize == 128


class ClassDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("n", C.c_int32), ("phi", C.c_int32), ("phi_n", C.c_int32), ("dim", C.c_int32),
        ("planner_mode", C.c_int32), ("use_sfc", C.c_int32), ("row_format", C.c_int32),
        ("dt", C.c_double), ("control_input_weight", C.c_double), ("terminal_weight", C.c_double),
        ("communication_range", C.c_double), ("world_min", C.c_double * 3), ("world_max", C.c_double * 3),
        ("max_iter", C.c_int32), ("precision", C.c_int32), ("tol", C.c_double), ("warm_start", C.c_int32), ("active_set", C.c_int32),
    ]


class Work(C.Structure):  # lscqp_work
    _fields_ = [("flops_fixed", C.c_double), ("flops_per_iteration", C.c_double), ("flops_last_pass", C.c_double),
                ("f64_insts_per_iteration", C.c_double), ("valu_insts_per_iteration", C.c_double), ("lds_insts_per_iteration", C.c_double),
                ("valu_insts_fixed", C.c_double), ("wavefronts", C.c_int32), ("nslot", C.c_int32), ("max_obstacles", C.c_int32),
                ("lds_bytes", C.c_int32), ("kernel", C.c_char * 96)]


class LscqpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lscqp error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load liblscqp.so.  Raises if it has not been built (python -m lsc_dr_planner_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("liblscqp.so not built: run `python lsc_dr_planner_amd/build.py` "
                              "(there is no CPU fallback)")
        # torch bundles its own libamdhip64 / libhsa-runtime64.  If liblscqp.so (linked against /opt/rocm's copy) is
        # loaded first, the process ends up with TWO HIP runtimes and the one that initialises second sees no device.
        # Importing torch first makes the loader resolve liblscqp.so's libamdhip64.so.7 to the copy torch already
        # loaded, so device pointers, streams and events are shared.  (The C++ shim has no torch and uses /opt/rocm.)
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.lscqp_create.restype = C.c_int
        L.lscqp_create.argtypes = [C.POINTER(ClassDesc), C.POINTER(vp)]
        L.lscqp_update.restype = C.c_int
        L.lscqp_update.argtypes = [vp, C.POINTER(ClassDesc)]
        L.lscqp_destroy.restype = C.c_int
        L.lscqp_destroy.argtypes = [vp]
        L.lscqp_num_variables.restype = C.c_int
        L.lscqp_num_variables.argtypes = [vp]
        L.lscqp_num_inequalities.restype = C.c_int
        L.lscqp_num_inequalities.argtypes = [vp, C.c_int32]
        L.lscqp_algorithmic_bytes.restype = C.c_int64
        L.lscqp_algorithmic_bytes.argtypes = [vp, C.c_int32]
        L.lscqp_solve_batch.restype = C.c_int
        L.lscqp_solve_batch.argtypes = [vp, C.c_int64] + [vp] * 9
        L.lscqp_solve_batch_device.restype = C.c_int
        L.lscqp_solve_batch_device.argtypes = [vp, C.c_int64, C.c_int32] + [vp] * 10
        L.lscqp_solve_batch_device_ex.restype = C.c_int
        L.lscqp_solve_batch_device_ex.argtypes = [vp, C.c_int64, C.c_int32] + [vp] * 9 + [C.c_int32, vp]
        L.lscqp_solve_batch_device_ordered.restype = C.c_int
        L.lscqp_solve_batch_device_ordered.argtypes = [vp, C.c_int64, C.c_int32] + [vp] * 9 + [C.c_int32, vp, vp]
        L.lscqp_launch_capacity.restype = C.c_int64
        L.lscqp_launch_capacity.argtypes = [vp, C.c_int64, C.c_int32]
        L.lscqp_order_by_cost_device.restype = C.c_int
        L.lscqp_order_by_cost_device.argtypes = [C.c_int64, vp, vp, vp]
        L.lscqp_construct_sfc_device_ordered.restype = C.c_int
        L.lscqp_construct_sfc_device_ordered.argtypes = [vp, vp, C.c_int32, C.c_int64] + [vp] * 7
        L.lscqp_order_by_work_device.restype = C.c_int
        L.lscqp_order_by_work_device.argtypes = [C.c_int64, vp, vp, vp]
        L.lscqp_solve_batch_stream.restype = C.c_int
        L.lscqp_solve_batch_stream.argtypes = [vp, C.c_int64] + [vp] * 10
        for f in ("lscqp_num_segments", "lscqp_uses_sfc", "lscqp_row_bytes", "lscqp_max_obstacles"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp]
        L.lscqp_comm_create.restype = C.c_int
        L.lscqp_comm_create.argtypes = [C.c_int32, vp, C.POINTER(vp)]
        L.lscqp_comm_destroy.restype = None
        L.lscqp_comm_destroy.argtypes = [vp]
        L.lscqp_comm_size.restype = C.c_int32
        L.lscqp_comm_size.argtypes = [vp]
        L.lscqp_comm_device.restype = C.c_int32
        L.lscqp_comm_device.argtypes = [vp, C.c_int32]
        L.lscqp_comm_stream.restype = vp
        L.lscqp_comm_stream.argtypes = [vp, C.c_int32]
        L.lscqp_comm_backend.restype = C.c_char_p
        L.lscqp_comm_backend.argtypes = [vp]
        L.lscqp_comm_set_min_agents_per_device.restype = C.c_int
        L.lscqp_comm_set_min_agents_per_device.argtypes = [vp, C.c_int64]
        L.lscqp_comm_devices_for.restype = C.c_int32
        L.lscqp_comm_devices_for.argtypes = [vp, C.c_int64]
        L.lscqp_comm_devices_for_class.restype = C.c_int32
        L.lscqp_comm_devices_for_class.argtypes = [vp, vp, C.c_int64, C.c_int32]
        L.lscqp_device_fill.restype = C.c_int64
        L.lscqp_device_fill.argtypes = [vp, C.c_int64, C.c_int32]
        L.lscqp_comm_shard.restype = C.c_int
        L.lscqp_comm_shard.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, vp, vp]
        L.lscqp_shard_range.restype = C.c_int
        L.lscqp_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp]
        L.lscqp_exchange_schedule.restype = C.c_int
        L.lscqp_exchange_schedule.argtypes = [C.c_int64, C.c_int32, vp, vp, C.c_int64, vp, C.c_int32, vp]
        L.lscqp_exchange_schedule_padded.restype = C.c_int
        L.lscqp_exchange_schedule_padded.argtypes = [C.c_int64, C.c_int32, vp, vp, C.c_int64, C.c_int64, vp, C.c_int32, vp]
        L.lscqp_comm_synchronize.restype = C.c_int
        L.lscqp_comm_synchronize.argtypes = [vp]
        L.lscqp_solve_batch_sharded.restype = C.c_int
        L.lscqp_solve_batch_sharded.argtypes = [vp, vp, C.c_int64] + [vp] * 10
        L.lscqp_solve_batch_sharded_device.restype = C.c_int
        L.lscqp_solve_batch_sharded_device.argtypes = [vp, vp, vp, C.c_int32] + [vp] * 9 + [C.c_int32]
        L.lscqp_allgather.restype = C.c_int
        L.lscqp_allgather.argtypes = [vp, vp, vp, C.c_int64]
        L.lscqp_generate_lsc_device.restype = C.c_int
        L.lscqp_generate_lsc_device.argtypes = [vp, C.c_int64, C.c_int32, C.c_int64] + [vp] * 7
        L.lscqp_generate_constraints_device.restype = C.c_int
        L.lscqp_generate_constraints_device.argtypes = [vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64] + [vp] * 7
        L.lscqp_generate_constraints_device_ex.restype = C.c_int
        L.lscqp_generate_constraints_device_ex.argtypes = [vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64] + [vp] * 6 + [C.c_int32, C.c_int32, vp]
        L.lscqp_generate_lsc_obstacles_device.restype = C.c_int
        L.lscqp_generate_lsc_obstacles_device.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int64] + [vp] * 7 + [C.c_int32, C.c_int32, vp]
        L.lscqp_shift_traj_partial_device.restype = C.c_int
        L.lscqp_shift_traj_partial_device.argtypes = [vp, C.c_int64, C.c_double, C.c_double, vp, vp, vp]
        L.lscqp_select_neighbours_device.restype = C.c_int
        L.lscqp_select_neighbours_device.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double, vp, vp, vp, vp]
        L.lscqp_shift_traj_device.restype = C.c_int
        L.lscqp_shift_traj_device.argtypes = [vp, C.c_int64, C.c_int32, C.c_double, vp, vp, vp]
        L.lscqp_generate_lsc_bytes.restype = C.c_int64
        L.lscqp_generate_lsc_bytes.argtypes = [vp, C.c_int64, C.c_int32, C.c_int64]
        L.lscqp_optimize_goal_device.restype = C.c_int
        L.lscqp_optimize_goal_device.argtypes = [vp, C.c_int64] + [vp] * 6
        L.lscqp_optimize_goal.restype = C.c_int
        L.lscqp_optimize_goal.argtypes = [vp, C.c_int64] + [vp] * 5
        L.lscqp_safety_metrics_device.restype = C.c_int
        L.lscqp_safety_metrics_device.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double] + [vp] * 6
        L.lscqp_safety_obstacles_device.restype = C.c_int
        L.lscqp_safety_obstacles_device.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double, vp, vp, vp, C.c_int32, vp, vp, vp]
        L.lscqp_map_create.restype = C.c_int
        L.lscqp_map_create.argtypes = [vp, C.c_int64, vp, vp, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        L.lscqp_map_create_from_csv.restype = C.c_int
        L.lscqp_map_create_from_csv.argtypes = [C.c_char_p, vp, vp, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        L.lscqp_map_destroy.restype = None
        L.lscqp_map_destroy.argtypes = [vp]
        L.lscqp_map_info.restype = C.c_int
        L.lscqp_map_info.argtypes = [vp, vp, vp]
        L.lscqp_map_download.restype = C.c_int
        L.lscqp_map_download.argtypes = [vp, vp, vp]
        L.lscqp_construct_sfc.restype = C.c_int
        L.lscqp_construct_sfc.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp]
        L.lscqp_construct_sfc_device.restype = C.c_int
        L.lscqp_construct_sfc_device.argtypes = [vp, vp, C.c_int32, C.c_int64] + [vp] * 5
        L.lscqp_validate_step_device.restype = C.c_int
        L.lscqp_validate_step_device.argtypes = [vp, C.c_int64, C.c_double, C.c_double] + [vp] * 6
        L.lscqp_plan_create.restype = C.c_int
        L.lscqp_plan_create.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_void_p)]
        L.lscqp_plan_destroy.restype = None
        L.lscqp_plan_destroy.argtypes = [vp]
        L.lscqp_plan_reset.restype = C.c_int
        L.lscqp_plan_reset.argtypes = [vp, vp, vp]
        L.lscqp_plan_buffer.restype = C.c_void_p
        L.lscqp_plan_buffer.argtypes = [vp, C.c_int32, vp]
        L.lscqp_plan_upload.restype = C.c_int
        L.lscqp_plan_upload.argtypes = [vp, C.c_int32, vp, C.c_uint64, C.c_uint64]
        L.lscqp_plan_download.restype = C.c_int
        L.lscqp_plan_download.argtypes = [vp, C.c_int32, vp, C.c_uint64, C.c_uint64]
        L.lscqp_plan_step.restype = C.c_int
        L.lscqp_plan_step.argtypes = [vp, vp]
        L.lscqp_plan_step_graph.restype = C.c_int
        L.lscqp_plan_step_graph.argtypes = [vp, vp]
        L.lscqp_plan_graph_nodes.restype = C.c_int64
        L.lscqp_plan_graph_nodes.argtypes = [vp]
        L.lscqp_plan_group_step.restype = C.c_int
        L.lscqp_plan_group_step.argtypes = [vp, vp, C.c_int32]
        L.lscqp_last_error.restype = C.c_char_p
        L.lscqp_version.restype = C.c_char_p
        L.lscqp_instance_work.restype = C.c_int
        L.lscqp_instance_work.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(Work)]
        L.lscqp_diagnose.restype = C.c_int
        L.lscqp_diagnose.argtypes = [vp, C.c_int64] + [vp] * 5 + [C.c_double, vp]
        L.lscqp_diagnose_device.restype = C.c_int
        L.lscqp_diagnose_device.argtypes = [vp, C.c_int64] + [vp] * 5 + [C.c_double, vp, vp]
        L.lscqp_dump_instance.restype = C.c_int
        L.lscqp_dump_instance.argtypes = [vp, vp, vp, vp, C.c_char_p]
        L.lscqp_row_family_name.restype = C.c_char_p
        L.lscqp_row_family_name.argtypes = [C.c_int32]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["lscqp_create", "lscqp_update", "lscqp_destroy", "lscqp_num_variables", "lscqp_num_inequalities",
                    "lscqp_algorithmic_bytes", "lscqp_solve_batch", "lscqp_solve_batch_stream", "lscqp_solve_batch_device", "lscqp_solve_batch_device_ex", "lscqp_solve_batch_device_ordered", "lscqp_order_by_work_device", "lscqp_launch_capacity", "lscqp_order_by_cost_device", "lscqp_construct_sfc_device_ordered",
                    "lscqp_num_segments", "lscqp_uses_sfc", "lscqp_row_bytes", "lscqp_max_obstacles", "lscqp_prepare_device", "lscqp_comm_prepare", "lscqp_comm_create", "lscqp_comm_destroy", "lscqp_comm_size",
                    "lscqp_comm_device", "lscqp_comm_stream", "lscqp_comm_backend", "lscqp_comm_set_min_agents_per_device",
                    "lscqp_comm_devices_for", "lscqp_comm_devices_for_class", "lscqp_device_fill", "lscqp_comm_shard", "lscqp_shard_range", "lscqp_exchange_schedule", "lscqp_exchange_schedule_padded", "lscqp_comm_synchronize", "lscqp_solve_batch_sharded",
                    "lscqp_solve_batch_sharded_device", "lscqp_allgather", "lscqp_generate_lsc_device", "lscqp_select_neighbours_device", "lscqp_generate_constraints_device",
                    "lscqp_shift_traj_device", "lscqp_shift_traj_partial_device", "lscqp_generate_constraints_device_ex",
                    "lscqp_generate_lsc_obstacles_device", "lscqp_generate_lsc_bytes", "lscqp_optimize_goal_device", "lscqp_optimize_goal", "lscqp_validate_step_device", "lscqp_map_create", "lscqp_map_create_from_csv", "lscqp_map_destroy", "lscqp_map_info",
                    "lscqp_map_download", "lscqp_map_prepare", "lscqp_construct_sfc_device", "lscqp_construct_sfc", "lscqp_safety_metrics_device", "lscqp_safety_obstacles_device",
                    "lscqp_plan_create", "lscqp_plan_destroy", "lscqp_plan_reset", "lscqp_plan_buffer", "lscqp_plan_upload", "lscqp_plan_download",
                    "lscqp_plan_step", "lscqp_plan_step_graph", "lscqp_plan_graph_nodes", "lscqp_plan_group_step",
                    "lscqp_instance_work", "lscqp_diagnose", "lscqp_diagnose_device", "lscqp_dump_instance", "lscqp_row_family_name",
                    "lscqp_last_error", "lscqp_version"]


def make_desc(M=5, dim=3, dt=0.2, w_c=0.01, w_t=1.0, comm_range=3.0, planner_mode=PLANNER_LSC, use_sfc=True,
              world_min=(-5, -5, 0), world_max=(5, 5, 2.5), n=5, phi=3, phi_n=1, max_iter=0, tol=0.0, row_format=ROWS_F64,
              precision=PRECISION_F64, warm_start=0, active_set=0):
    d = ClassDesc()
    d.warm_start = warm_start
    d.active_set = active_set
    d.row_format = row_format
    d.precision = precision
    d.M, d.n, d.phi, d.phi_n, d.dim = M, n, phi, phi_n, dim
    d.planner_mode, d.use_sfc = planner_mode, int(use_sfc)
    d.dt, d.control_input_weight, d.terminal_weight, d.communication_range = dt, w_c, w_t, comm_range
    for k in range(3):
        d.world_min[k] = float(world_min[k])
        d.world_max[k] = float(world_max[k])
    d.max_iter, d.tol = max_iter, tol
    return d


def pack_rows(lsc):
    """Reference LSC records (p, nrm, d) -> packed rows (nx, ny, nz, b = d + nrm.p); include/lscqp.h lscqp_row."""
    out = np.zeros(lsc.shape, ROW_DTYPE)
    out["nx"], out["ny"], out["nz"] = lsc["nrm"][..., 0], lsc["nrm"][..., 1], lsc["nrm"][..., 2]
    out["b"] = lsc["d"] + (lsc["nrm"] * lsc["p"]).sum(-1)
    return out


class WorldMap:
    """The voxel map of the corridor construction (lscqp_map): world boxes (n, 6) = centre xyz, size xyz -- the rows of the
    reference's world CSV -- or the CSV file itself; lives in HBM."""

    def __init__(self, boxes=None, world_min=(-5, -5, 0), world_max=(5, 5, 2.5), resolution=0.1, max_dist=1.0, csv_path=None):
        wmin = np.ascontiguousarray(world_min, dtype=np.float64)
        wmax = np.ascontiguousarray(world_max, dtype=np.float64)
        h = C.c_void_p()
        if csv_path is not None:
            rc = lib().lscqp_map_create_from_csv(os.fsencode(csv_path), wmin.ctypes.data_as(C.c_void_p), wmax.ctypes.data_as(C.c_void_p),
                                                 float(resolution), float(max_dist), C.byref(h))
        else:
            b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
            rc = lib().lscqp_map_create(b.ctypes.data_as(C.c_void_p), b.shape[0], wmin.ctypes.data_as(C.c_void_p),
                                        wmax.ctypes.data_as(C.c_void_p), float(resolution), float(max_dist), C.byref(h))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        self._h = h
        dims, key0 = np.zeros(3, np.int32), np.zeros(3, np.int32)
        lib().lscqp_map_info(self._h, dims.ctypes.data_as(C.c_void_p), key0.ctypes.data_as(C.c_void_p))
        self.dims, self.key0 = dims, key0

    def prepare(self, max_radius):
        """lscqp_map_prepare: the free-space table that lets the corridor kernel pass tests in open space without sampling."""
        lib().lscqp_map_prepare.restype = C.c_int
        lib().lscqp_map_prepare.argtypes = [C.c_void_p, C.c_double]
        rc = lib().lscqp_map_prepare(self._h, float(max_radius))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def download(self):
        """(occ uint8, nearest int32), both shaped (dims[2], dims[1], dims[0])."""
        shape = (int(self.dims[2]), int(self.dims[1]), int(self.dims[0]))
        occ, near = np.zeros(shape, np.uint8), np.zeros(shape, np.int32)
        rc = lib().lscqp_map_download(self._h, occ.ctypes.data_as(C.c_void_p), near.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return occ, near

    def close(self):
        if self._h:
            lib().lscqp_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


AGENT_PARAM_DTYPE = np.dtype([("radius", "f8"), ("downwash", "f8"), ("max_vel", "f8", 3), ("max_acc", "f8", 3), ("nominal_velocity", "f8")])


class PlanDesc(C.Structure):  # lscqp_plan_desc
    _fields_ = [("n_agents", C.c_int64), ("n_total", C.c_int64), ("first_agent", C.c_int64), ("n_obs", C.c_int32), ("constraint_mode", C.c_int32),
                ("sfc_mode", C.c_int32), ("optimize_goal", C.c_int32), ("closed_loop", C.c_int32), ("safety_samples", C.c_int32),
                ("time_step", C.c_double), ("z_2d", C.c_double), ("record_time_step", C.c_double), ("tight_warm_start", C.c_int32),
                ("prediction_mode", C.c_int32), ("initial_traj_mode", C.c_int32), ("reserved_", C.c_int32), ("reset_threshold", C.c_double)]


TRAJ_FROM_PREVIOUS_SOLUTION, TRAJ_FROM_POSITION, TRAJ_FROM_VELOCITY = 0, 1, 2


(PLAN_STATE, PLAN_WAYPOINT, PLAN_PLAN, PLAN_GOAL, PLAN_HEADER, PLAN_ROWS, PLAN_SFC, PLAN_STATUS, PLAN_GOAL_STATUS, PLAN_SFC_STATUS, PLAN_VALID,
 PLAN_IN_RANGE, PLAN_NEXT_STATE, PLAN_OBJECTIVE, PLAN_INFO, PLAN_SAFETY) = range(16)


class Plan:
    """lscqp_plan: one replan of a batch of agents as one chain of device work (include/lscqp.h, "the caller of the path"), eager
    (`step`) or through a captured hipGraph (`step_graph`).  Buffers are addressed by the PLAN_* constants; `get` / `put` copy
    synchronously, `pointer` returns the device address."""

    _DT = {PLAN_STATE: np.float64, PLAN_WAYPOINT: np.float64, PLAN_PLAN: np.float64, PLAN_GOAL: np.float64, PLAN_STATUS: np.int32,
           PLAN_GOAL_STATUS: np.int32, PLAN_SFC_STATUS: np.int32, PLAN_VALID: np.int32, PLAN_IN_RANGE: np.int32, PLAN_NEXT_STATE: np.float64,
           PLAN_OBJECTIVE: np.float64}

    def __init__(self, solver, world_map, n_agents, n_obs, agents, n_total=None, first_agent=0, constraint_mode=1, sfc_mode=1,
                 optimize_goal=True, closed_loop=False, time_step=None, z_2d=1.0, safety_samples=0, record_time_step=0.1, tight_warm_start=False,
                 prediction_mode=TRAJ_FROM_PREVIOUS_SOLUTION, initial_traj_mode=TRAJ_FROM_PREVIOUS_SOLUTION, reset_threshold=0.1):
        self._p = None
        n_total = n_agents if n_total is None else n_total
        d = PlanDesc()
        d.n_agents, d.n_total, d.first_agent, d.n_obs = n_agents, n_total, first_agent, n_obs
        d.constraint_mode, d.sfc_mode, d.optimize_goal, d.closed_loop = constraint_mode, sfc_mode, int(optimize_goal), int(closed_loop)
        d.time_step = float(solver.desc.dt if time_step is None else time_step)
        d.z_2d = float(z_2d)
        d.safety_samples, d.record_time_step = int(safety_samples), float(record_time_step)
        d.tight_warm_start = int(tight_warm_start)
        d.prediction_mode, d.initial_traj_mode, d.reset_threshold = int(prediction_mode), int(initial_traj_mode), float(reset_threshold)
        ag = np.ascontiguousarray(agents, dtype=AGENT_PARAM_DTYPE)
        if ag.shape != (n_total,):
            raise ValueError("agents: one AGENT_PARAM_DTYPE record per agent of the mission")
        h = C.c_void_p()
        rc = lib().lscqp_plan_create(solver._h, world_map._h if world_map is not None else None, C.byref(d), ag.ctypes.data_as(C.c_void_p), C.byref(h))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        self._p, self._solver, self._map = h, solver, world_map  # (keeps the solver and the map alive)
        self.n_agents, self.n_total, self.first_agent, self.n_obs, self.M, self.nv = n_agents, n_total, first_agent, n_obs, solver.desc.M, solver.nv
        self._dt = dict(self._DT)
        self._dt.update({PLAN_HEADER: HEADER_DTYPE, PLAN_ROWS: ROW_DTYPE, PLAN_SFC: BOX_DTYPE, PLAN_INFO: INFO_DTYPE, PLAN_SAFETY: SAFETY_DTYPE})

    def close(self):
        if self._p:
            lib().lscqp_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def reset(self, start_positions, goal_points=None):
        sp = np.ascontiguousarray(start_positions, dtype=np.float64).reshape(self.n_total, 3)
        gp = None if goal_points is None else np.ascontiguousarray(goal_points, dtype=np.float64).reshape(self.n_total, 3)
        self._check(lib().lscqp_plan_reset(self._p, sp.ctypes.data_as(C.c_void_p), None if gp is None else gp.ctypes.data_as(C.c_void_p)))

    def pointer(self, which):
        nb = C.c_uint64()
        ptr = lib().lscqp_plan_buffer(self._p, which, C.byref(nb))
        return ptr, nb.value

    def get(self, which):
        _, nb = self.pointer(which)
        out = np.zeros(nb // np.dtype(self._dt[which]).itemsize, dtype=self._dt[which])
        self._check(lib().lscqp_plan_download(self._p, which, out.ctypes.data_as(C.c_void_p), 0, nb))
        return out

    def put(self, which, array, first=0):
        """array: entries [first, first + len) of the buffer, in units of the buffer's record (an agent's 9 doubles, ...)."""
        a = np.ascontiguousarray(array, dtype=self._dt[which])
        _, nb = self.pointer(which)
        per = {PLAN_STATE: 72, PLAN_WAYPOINT: 24, PLAN_GOAL: 24, PLAN_PLAN: 8 * self.nv, PLAN_SFC: 48 * self.M}.get(which)
        if per is None:
            raise ValueError("not an input buffer")
        self._check(lib().lscqp_plan_upload(self._p, which, a.ctypes.data_as(C.c_void_p), first * per, a.nbytes))

    def tensor(self, which):
        """Zero-copy torch view of a float64 buffer of the plan (PLAN_PLAN, PLAN_STATE, PLAN_GOAL, ...): what
        sharding.exchange_plan_buffers exchanges between the ranks of a torch.distributed job."""
        import torch

        if np.dtype(self._dt[which]) != np.float64:
            raise ValueError("not a float64 buffer")
        ptr, nb = self.pointer(which)

        class _View:
            __cuda_array_interface__ = dict(shape=(nb // 8,), typestr="<f8", data=(ptr, False), version=2)

        return torch.as_tensor(_View(), device="cuda")

    def step(self, stream=None, graph=False):
        if getattr(self, "_solver", None) is not None:
            self._solver._sync_knobs()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._check((lib().lscqp_plan_step_graph if graph else lib().lscqp_plan_step)(self._p, sp))

    def graph_nodes(self):
        return int(lib().lscqp_plan_graph_nodes(self._p))


def shard_range(n, n_used, g):
    """lscqp_shard_range: block [first, first + count) of device g when n agents are spread over n_used devices."""
    f, c = C.c_int64(), C.c_int64()
    rc = lib().lscqp_shard_range(n, n_used, g, C.byref(f), C.byref(c))
    if rc != OK:
        raise LscqpError(rc, lib().lscqp_last_error().decode())
    return f.value, c.value


XCHG_ALLGATHER, XCHG_BROADCAST = 0, 1
EXCHANGE_OP_DTYPE = np.dtype([("kind", "<i4"), ("root", "<i4"), ("offset", "<i8"), ("count", "<i8")])


PLAN_EXCHANGE_PAD = 64  # LSCQP_PLAN_EXCHANGE_PAD


def exchange_schedule(n_total, first, count, per=1, pad_agents=None):
    """lscqp_exchange_schedule[_padded]: the collective operations of the exchange that follows a sharded replan (what lscqp_plan_group_step
    runs over RCCL), for blocks [first[g], first[g] + count[g]) of n_total agents, `per` doubles per agent; pad_agents: agents of room
    behind the mission in every device's buffer (None: the unpadded entry).  No device needed."""
    first, count = np.ascontiguousarray(first, dtype=np.int64), np.ascontiguousarray(count, dtype=np.int64)
    ops = np.zeros(max(len(first), 1), EXCHANGE_OP_DTYPE)
    n = C.c_int32()
    if pad_agents is not None:
        rc = lib().lscqp_exchange_schedule_padded(int(n_total), len(first), first.ctypes.data, count.ctypes.data, int(per), int(pad_agents), ops.ctypes.data, len(ops), C.byref(n))
    else:
        rc = lib().lscqp_exchange_schedule(int(n_total), len(first), first.ctypes.data, count.ctypes.data, int(per), ops.ctypes.data, len(ops), C.byref(n))
    if rc != OK:
        raise LscqpError(rc, lib().lscqp_last_error().decode())
    return ops[: n.value]


class Comm:
    """lscqp_comm: one host process driving the GPUs of a node (private stream + staging pool + RCCL communicator per device)."""

    def __init__(self, n_devices=0, device_ids=None):
        self._h = C.c_void_p()
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        rc = lib().lscqp_comm_create(int(n_devices), None if ids is None else ids.ctypes.data_as(C.c_void_p), C.byref(self._h))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        self.size = lib().lscqp_comm_size(self._h)
        self.backend = lib().lscqp_comm_backend(self._h).decode()

    def close(self):
        if self._h:
            lib().lscqp_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_min_agents_per_device(self, n):
        rc = lib().lscqp_comm_set_min_agents_per_device(self._h, int(n))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def devices_for(self, n):
        return lib().lscqp_comm_devices_for(self._h, int(n))

    def devices_for_class(self, solver, n, n_obs_max):
        solver._sync_knobs()
        return lib().lscqp_comm_devices_for_class(self._h, solver._h, int(n), int(n_obs_max))

    def prepare(self, solver):
        """lscqp_comm_prepare: the solver's active-set tables on every device of the communicator."""
        rc = lib().lscqp_comm_prepare(self._h, solver._h)
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def stream(self, g):
        return lib().lscqp_comm_stream(self._h, g)

    def synchronize(self):
        rc = lib().lscqp_comm_synchronize(self._h)
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def plan_group_step(self, plans, graph=False):
        """lscqp_plan_group_step: plans[g] lives on device g and owns its block of the mission's agents; every plan's replan is
        enqueued on its device's stream, followed by the in-place RCCL exchange of the owners' plan / state / goal slices."""
        if len(plans) != self.size:
            raise ValueError("one plan per device of the communicator")
        arr = (C.c_void_p * self.size)(*[p._p.value for p in plans])
        rc = lib().lscqp_plan_group_step(self._h, arr, int(bool(graph)))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def allgather(self, send, recv, count):
        """send / recv: lists of torch CUDA tensors, one per device (device g contributes send[g][:count] doubles and
        receives size * count); asynchronous on the communicator's streams."""
        vp = C.c_void_p * self.size
        ps = vp(*[t.data_ptr() for t in send])
        pr = vp(*[t.data_ptr() for t in recv])
        rc = lib().lscqp_allgather(self._h, ps, pr, int(count))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())


# The library reads its environment when a handle is CREATED and never afterwards (csrc/lscqp_api.hip: Knobs).  The test suite and bench.py
# flip these switches between launches of one process, on live handles: this wrapper (test and bench plumbing, not the product) notices a change
# of the process environment and tells the handle to re-read it through the library-internal lscqp_debug_reload_knobs_.
_KNOB_ENV = ("LSCQP_FORCE_GENERIC", "LSCQP_WAVES", "LSCQP_ACTIVE_SET", "LSCQP_ACTIVE_SET_NOW", "LSCQP_CHECK_ORDER", "LSCQP_NO_QUEUE", "LSCQP_DEFER_BEHIND", "LSCQP_ZERO_COPY_BYTES")


def _knob_env():
    g = os.environ.get
    return tuple(g(k) for k in _KNOB_ENV)


class Solver:
    def __init__(self, desc):
        self.desc = desc
        self._h = C.c_void_p()
        self._knob_seen = _knob_env()
        rc = lib().lscqp_create(C.byref(desc), C.byref(self._h))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        self.nv = lib().lscqp_num_variables(self._h)
        self.M, self.dim, self.P = desc.M, desc.dim, desc.M * 6

    def _sync_knobs(self):
        now = _knob_env()
        if now != self._knob_seen:
            self._knob_seen = now
            lib().lscqp_debug_reload_knobs_(self._h)

    def set_knob(self, name, value):
        """lscqp_debug_set_knob_ (library-internal): one of the handle's development switches by name, e.g. the launch-shape overrides of
        the dual active-set phase (das_threads, das_kmax, das_steps, das_cache, das_stage, das_screen, das_loop; -1 = the policy's value)."""
        rc = lib().lscqp_debug_set_knob_(self._h, name.encode(), int(value))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def prepare_device(self):
        rc = lib().lscqp_prepare_device(self._h)
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def bind_device(self, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_status, d_info=None, stream=None, d_x_init=None, retry=False,
                    d_order=None):
        """solve_device with every argument converted ONCE: returns a zero-argument callable that enqueues the same
        lscqp_solve_batch_device_ordered call on the same stream each time it is called (a timed loop then pays for the C entry, not for
        building eleven ctypes pointers per step).  The tensors must stay alive and in place."""
        import torch

        self._sync_knobs()
        s = stream if stream is not None else torch.cuda.current_stream()

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        fn = lib().lscqp_solve_batch_device_ordered
        args = (self._h, n, n_obs_max, p(d_hdr), p(d_rows), p(d_off), p(d_sfc), p(d_x_init), p(d_x), p(d_obj), p(d_status), p(d_info), int(retry),
                p(d_order), C.c_void_p(s.cuda_stream))
        keep = (d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_status, d_info, d_x_init, d_order, s)

        def call(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc != OK:
                raise LscqpError(rc, lib().lscqp_last_error().decode())

        return call

    def rows_in_format(self, rows):
        """Packed rows (ROW_DTYPE or ROW_F32_DTYPE) in the handle's storage format; fp64 rows are rounded to float32 for
        a ROWS_F32 handle (what a producer writing that format would store)."""
        rows = np.asarray(rows)
        want = ROW_F32_DTYPE if self.desc.row_format == ROWS_F32 else ROW_DTYPE
        if rows.dtype != want:
            src = np.ascontiguousarray(rows, dtype=ROW_DTYPE if rows.dtype.names is None else rows.dtype).reshape(-1)
            out = np.zeros(src.shape, want)
            for f in ("nx", "ny", "nz", "b"):
                out[f] = src[f]
            rows = out
        return np.ascontiguousarray(rows).reshape(-1)

    def close(self):
        if self._h:
            lib().lscqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, desc):
        rc = lib().lscqp_update(self._h, C.byref(desc))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        self.desc = desc

    def max_obstacles(self):
        return lib().lscqp_max_obstacles(self._h)

    def instance_work(self, n, n_obs_max):
        """lscqp_instance_work: work counters (from the machine code) of the kernel instance a launch of n QPs would select."""
        self._sync_knobs()
        w = Work()
        rc = lib().lscqp_instance_work(self._h, int(n), int(n_obs_max), C.byref(w))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return {f: (getattr(w, f).decode() if f == "kernel" else getattr(w, f)) for f, _ in Work._fields_}

    def algorithmic_bytes(self, n_obs):
        return lib().lscqp_algorithmic_bytes(self._h, n_obs)

    def num_inequalities(self, n_obs):
        return lib().lscqp_num_inequalities(self._h, n_obs)

    # ---- host-pointer call (numpy) --------------------------------------------------------------------
    def solve_host(self, hdr, rows=None, row_offsets=None, sfc=None, want_info=True, x_init=None):
        """x_init: (n, nv) initial trajectories (TrajOptimizer::solve's initial_traj) as the primal start, or None."""
        self._sync_knobs()
        n = len(hdr)
        hdr = np.ascontiguousarray(hdr, dtype=HEADER_DTYPE)
        x = np.zeros((n, self.nv))
        obj = np.zeros(n)
        status = np.full(n, -1, dtype=np.int32)
        info = np.zeros(n, INFO_DTYPE) if want_info else None
        if rows is not None:
            rows = self.rows_in_format(rows)
            row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
            assert len(row_offsets) == n + 1
        if sfc is not None:
            sfc = np.ascontiguousarray(sfc, dtype=BOX_DTYPE).reshape(-1)

        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        if x_init is not None:
            x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(n, self.nv)
        rc = lib().lscqp_solve_batch(self._h, n, p(hdr), p(rows), p(row_offsets), p(sfc), p(x_init), p(x), p(obj), p(status), p(info))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return dict(x=x, obj=obj, status=status, info=info)

    def solve_sharded(self, comm, hdr, rows=None, row_offsets=None, sfc=None, want_info=True, x_init=None):
        """lscqp_solve_batch_sharded: the host-pointer call over the devices of `comm`; returns solve_host's dict + devices_used."""
        self._sync_knobs()
        n = len(hdr)
        hdr = np.ascontiguousarray(hdr, dtype=HEADER_DTYPE)
        x = np.zeros((n, self.nv))
        obj = np.zeros(n)
        status = np.full(n, -1, dtype=np.int32)
        info = np.zeros(n, INFO_DTYPE) if want_info else None
        if rows is not None:
            rows = self.rows_in_format(rows)
            row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
        if sfc is not None:
            sfc = np.ascontiguousarray(sfc, dtype=BOX_DTYPE).reshape(-1)
        if x_init is not None:
            x_init = np.ascontiguousarray(x_init, dtype=np.float64).reshape(n, self.nv)

        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        used = C.c_int32(0)
        rc = lib().lscqp_solve_batch_sharded(self._h, comm._h, n, p(hdr), p(rows), p(row_offsets), p(sfc), p(x_init), p(x), p(obj),
                                             p(status), p(info), C.cast(C.byref(used), C.c_void_p))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return dict(x=x, obj=obj, status=status, info=info, devices_used=used.value)

    # ---- failure diagnostics (include/lscqp.h) -------------------------------------------------------------
    def diagnose_host(self, hdr, rows, row_offsets, sfc, x, tol=1e-9):
        """lscqp_diagnose: every row of the reference's model evaluated on the trajectories x (n, nv) -> DIAG_DTYPE[n]."""
        n = len(hdr)
        hdr = np.ascontiguousarray(hdr, dtype=HEADER_DTYPE)
        out = np.zeros(n, DIAG_DTYPE)
        if rows is not None:
            rows = self.rows_in_format(rows)
            row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
        if sfc is not None:
            sfc = np.ascontiguousarray(sfc, dtype=BOX_DTYPE).reshape(-1)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(n, self.nv)

        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        rc = lib().lscqp_diagnose(self._h, n, p(hdr), p(rows), p(row_offsets), p(sfc), p(x), float(tol), p(out))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return out

    def dump_instance(self, hdr_one, rows_one, sfc_one, path):
        """lscqp_dump_instance: ONE instance as a CPLEX LP file (what cplex.exportModel writes in the reference); needs no device."""
        hdr = np.ascontiguousarray(hdr_one, dtype=HEADER_DTYPE).reshape(1)
        rows = None if rows_one is None else self.rows_in_format(rows_one)
        sfc = None if sfc_one is None else np.ascontiguousarray(sfc_one, dtype=BOX_DTYPE).reshape(-1)

        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        rc = lib().lscqp_dump_instance(self._h, p(hdr), p(rows), p(sfc), os.fsencode(path))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def solve_sharded_device(self, comm, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_status, d_info=None, d_x_init=None, retry=False):
        """lscqp_solve_batch_sharded_device: lists of per-device torch CUDA tensors (entry g lives on device g of `comm`), n[g] agents on
        device g; asynchronous on the communicator's streams (comm.synchronize() waits)."""
        self._sync_knobs()
        G = comm.size
        vp = C.c_void_p * G

        def arr(ts):
            return None if ts is None else vp(*[(None if t is None else t.data_ptr()) for t in ts])

        nn = (C.c_int64 * G)(*[int(v) for v in n])
        rc = lib().lscqp_solve_batch_sharded_device(self._h, comm._h, nn, int(n_obs_max), arr(d_hdr), arr(d_rows), arr(d_off), arr(d_sfc),
                                                    arr(d_x_init), arr(d_x), arr(d_obj), arr(d_status), arr(d_info), int(bool(retry)))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    # ---- device-pointer call (torch tensors hold the HBM buffers) --------------------------------------
    def solve_device(self, n, n_obs_max, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_status, d_info=None, stream=None,
                     d_x_init=None, retry=False, d_order=None):
        """All arguments are torch CUDA tensors (any dtype; only data_ptr() is used) or None.
        Asynchronous on `stream` (torch.cuda.Stream) or torch's current stream.  retry: lscqp_solve_batch_device_ex's second pass.
        d_order: int32 permutation of 0 .. n-1 (lscqp_solve_batch_device_ordered: the k-th slot of the launch solves instance d_order[k])."""
        import torch

        self._sync_knobs()

        s = stream if stream is not None else torch.cuda.current_stream()

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        rc = lib().lscqp_solve_batch_device_ordered(self._h, n, n_obs_max, p(d_hdr), p(d_rows), p(d_off), p(d_sfc), p(d_x_init), p(d_x),
                                                    p(d_obj), p(d_status), p(d_info), int(retry), p(d_order), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def device_fill(self, n, n_obs_max):
        """lscqp_device_fill: instances one device works on at once in the first kernel of a solve of this class."""
        self._sync_knobs()
        return lib().lscqp_device_fill(self._h, int(n), int(n_obs_max))

    def launch_capacity(self, n, n_obs_max):
        """lscqp_launch_capacity: instances of a launch of n the device works on at once (-1 without a device)."""
        self._sync_knobs()
        return int(lib().lscqp_launch_capacity(self._h, int(n), int(n_obs_max)))

    @staticmethod
    def order_by_cost_device(n, d_cost_prev, d_order_out, stream=None):
        """lscqp_order_by_cost_device: d_order_out (int32[n]) := agents by the cost (uint32) of their previous corridor, most expensive first."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_order_by_cost_device(int(n), C.c_void_p(d_cost_prev.data_ptr()), C.c_void_p(d_order_out.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    @staticmethod
    def order_by_work_device(n, d_info_prev, d_order_out, stream=None):
        """lscqp_order_by_work_device: d_order_out (int32[n]) := instances by the iterations of their previous solve, most first."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_order_by_work_device(int(n), C.c_void_p(d_info_prev.data_ptr()), C.c_void_p(d_order_out.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())


    # ---- GoalOptimizer::solve in closed form ----------------------------------------
    def optimize_goal_host(self, hdr, rows=None, row_offsets=None, sfc=None):
        """hdr["goal"] = current_goal_point on entry; returns (hdr with the optimised goal, status)."""
        n = len(hdr)
        hdr = np.ascontiguousarray(hdr, dtype=HEADER_DTYPE).copy()
        status = np.full(n, -1, dtype=np.int32)
        if rows is not None:
            rows = self.rows_in_format(rows)
            row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
        if sfc is not None:
            sfc = np.ascontiguousarray(sfc, dtype=BOX_DTYPE).reshape(-1)

        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        rc = lib().lscqp_optimize_goal(self._h, n, p(hdr), p(rows), p(row_offsets), p(sfc), p(status))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())
        return hdr, status

    def optimize_goal_device(self, n, d_hdr, d_rows, d_off, d_sfc, d_status, stream=None):
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        rc = lib().lscqp_optimize_goal_device(self._h, n, p(d_hdr), p(d_rows), p(d_off), p(d_sfc), p(d_status), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    # ---- isSolValid + getStateAt + doStep -------------------------------------------
    def safety_metrics_device(self, n_agents, first_agent, n_total, n_samples, record_time_step, d_x_all, d_radius, d_downwash, d_hdr,
                              d_out, z_2d=1.0, stream=None):
        """MultiSyncSimulator::update's safety ratio / excess ratios per local agent (SAFETY_DTYPE records in d_out)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        rc = lib().lscqp_safety_metrics_device(self._h, n_agents, first_agent, n_total, int(n_samples), float(record_time_step), float(z_2d),
                                               p(d_x_all), p(d_radius), p(d_downwash), p(d_hdr), p(d_out), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def safety_obstacles_device(self, n_agents, first_agent, n_total, n_samples, record_time_step, d_x_all, d_radius, d_downwash, n_obstacles,
                                d_obstacles, d_out, z_2d=1.0, stream=None):
        """MultiSyncSimulator::update's obstacle safety ratio per local agent (SAFETY_OBS_DTYPE records in d_out; d_obstacles: OBSTACLE_DTYPE)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)  # noqa: E731
        rc = lib().lscqp_safety_obstacles_device(self._h, n_agents, first_agent, n_total, int(n_samples), float(record_time_step), float(z_2d),
                                                 p(d_x_all), p(d_radius), p(d_downwash), int(n_obstacles), p(d_obstacles), p(d_out),
                                                 C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def construct_sfc_device(self, world_map, mode, n, d_points, d_radius, d_sfc, d_status, stream=None, d_order=None, d_cost=None):
        """Corridor update of n agents on the device (SFC_INIT / SFC_FROM_HULL / SFC_FROM_POINT, see include/lscqp.h).  d_order: int32
        permutation (workgroup k builds agent d_order[k]'s corridor); d_cost: uint32[n], every agent's cost of this launch."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)  # noqa: E731
        rc = lib().lscqp_construct_sfc_device_ordered(self._h, world_map._h, int(mode), n, p(d_points), p(d_radius), p(d_sfc), p(d_status),
                                                      p(d_order), p(d_cost), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def validate_step_device(self, n, time_step, d_x, d_hdr, d_sfc, d_valid, d_state, z_2d=1.0, stream=None):
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        rc = lib().lscqp_validate_step_device(self._h, n, float(time_step), float(z_2d), p(d_x), p(d_hdr), p(d_sfc), p(d_valid),
                                              p(d_state), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    # ---- the producer of the rows, device pointers -----------------------------------
    def generate_lsc_device(self, n_agents, n_obs, first_agent, d_traj, d_neighbours, d_radius, d_downwash, d_goal, d_rows,
                            stream=None):
        """TrajPlanner::generateLSC for agent obstacles, rows written in the layout solve_device consumes."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_generate_lsc_device(self._h, n_agents, n_obs, first_agent, C.c_void_p(d_traj.data_ptr()),
                                             C.c_void_p(d_neighbours.data_ptr()), C.c_void_p(d_radius.data_ptr()),
                                             C.c_void_p(d_downwash.data_ptr()), C.c_void_p(d_goal.data_ptr()),
                                             C.c_void_p(d_rows.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def generate_constraints_device(self, mode, n_agents, n_obs, first_agent, d_traj, d_neighbours, d_radius, d_downwash, d_goal_all,
                                    d_rows, stream=None):
        """generateLSC / generateCLSC / generateBVC (mode GEN_LSC / GEN_CLSC / GEN_BVC) on the device; d_goal_all holds the
        current goal point of every agent, indexed by global id (see include/lscqp.h)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_generate_constraints_device(self._h, int(mode), n_agents, n_obs, first_agent, C.c_void_p(d_traj.data_ptr()),
                                                     C.c_void_p(d_neighbours.data_ptr()), C.c_void_p(d_radius.data_ptr()),
                                                     C.c_void_p(d_downwash.data_ptr()), C.c_void_p(d_goal_all.data_ptr()),
                                                     C.c_void_p(d_rows.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def select_neighbours_device(self, n_agents, first_agent, n_total, n_obs, comm_range, d_positions, d_neighbours, d_count, stream=None):
        """broadcastMsgs' range filter on the device: neighbour ids (ascending, -1 padded) and the in-range counts."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_select_neighbours_device(self._h, n_agents, first_agent, n_total, int(n_obs), float(comm_range),
                                                  C.c_void_p(d_positions.data_ptr()), C.c_void_p(d_neighbours.data_ptr()),
                                                  C.c_void_p(d_count.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def shift_traj_device(self, n, d_x_prev, d_traj, z_2d=1.0, shift=1, stream=None):
        """initialTrajPlanningPrevSol: solver output [n][dim*M*6] -> initial trajectories [n][M][6][3] (float32 values)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_shift_traj_device(self._h, n, int(shift), float(z_2d), C.c_void_p(d_x_prev.data_ptr()), C.c_void_p(d_traj.data_ptr()),
                                           C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def shift_traj_partial_device(self, n, d_x_prev, d_traj, fraction, z_2d=1.0, stream=None):
        """multisim_time_step < dt: segment 0 := subSegment(fraction, 1) of the previous plan's, the others kept."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_shift_traj_partial_device(self._h, n, float(fraction), float(z_2d), C.c_void_p(d_x_prev.data_ptr()),
                                                   C.c_void_p(d_traj.data_ptr()), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def generate_constraints_device_ex(self, mode, n_agents, n_obs, first_agent, d_traj, d_neighbours, d_radius, d_downwash, d_goal_all,
                                       d_rows, n_obs_total, slot0, stream=None):
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_generate_constraints_device_ex(self._h, int(mode), n_agents, n_obs, first_agent, C.c_void_p(d_traj.data_ptr()),
                                                        C.c_void_p(d_neighbours.data_ptr()), C.c_void_p(d_radius.data_ptr()),
                                                        C.c_void_p(d_downwash.data_ptr()), C.c_void_p(d_goal_all.data_ptr()),
                                                        C.c_void_p(d_rows.data_ptr()), int(n_obs_total), int(slot0), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def generate_lsc_obstacles_device(self, param, n_agents, n_dyn, first_agent, d_traj, d_ids, d_obstacles, d_radius, d_goal, d_hdr, d_rows,
                                      n_obs_total, slot0, stream=None):
        """generateLSC for non-agent obstacles (param: ObstacleParam; d_obstacles: OBSTACLE_DTYPE table on the device)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        rc = lib().lscqp_generate_lsc_obstacles_device(self._h, C.cast(C.byref(param), C.c_void_p), n_agents, n_dyn, first_agent,
                                                       C.c_void_p(d_traj.data_ptr()), C.c_void_p(d_ids.data_ptr()),
                                                       C.c_void_p(d_obstacles.data_ptr()), C.c_void_p(d_radius.data_ptr()),
                                                       C.c_void_p(d_goal.data_ptr()), C.c_void_p(d_hdr.data_ptr()), C.c_void_p(d_rows.data_ptr()),
                                                       int(n_obs_total), int(slot0), C.c_void_p(s.cuda_stream))
        if rc != OK:
            raise LscqpError(rc, lib().lscqp_last_error().decode())

    def generate_lsc_bytes(self, n_agents, n_obs, n_total):
        return lib().lscqp_generate_lsc_bytes(self._h, n_agents, n_obs, n_total)


def x_init_from_swarm(build, dim):
    """synth.Swarm.build()["init"] (N, M, 6, 3) -> (N, dim*M*6) in the reference variable order (axis, segment, point)."""
    init = np.asarray(build["init"], dtype=np.float64)
    N = init.shape[0]
    return np.ascontiguousarray(init.transpose(0, 3, 1, 2)[:, :dim]).reshape(N, -1)


def batch_from_swarm(build, n_obs, M, vmax=1.0, amax=2.0, radius=0.15, nominal_velocity=1.0, terminal_segments=None):
    """Pack the output of synth.Swarm.build() into ABI arrays (hdr, rows, row_offsets, sfc)."""
    N = len(build["p0"])
    hdr = np.zeros(N, HEADER_DTYPE)
    for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
        hdr[f] = build[f]
    hdr["vmax"], hdr["amax"] = vmax, amax
    hdr["radius"], hdr["nominal_velocity"] = radius, nominal_velocity
    hdr["n_obs"] = n_obs
    hdr["terminal_segments"] = 0 if terminal_segments is None else terminal_segments
    rows = pack_rows(build["lsc"]).reshape(-1)
    off = (np.arange(N + 1, dtype=np.uint64) * np.uint64(n_obs * M * 6))
    sfc = np.zeros((N, M), BOX_DTYPE)
    sfc["bmin"], sfc["bmax"] = build["sfc"]["bmin"], build["sfc"]["bmax"]
    return hdr, rows, off, sfc
