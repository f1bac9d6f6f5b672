"""CPU tests of the C-ABI library: it loads, exports every symbol include/lscqp.h declares, its structs have the
documented sizes, argument validation mirrors the reference's exceptions, and WITHOUT a GPU it fails loudly
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lscqp.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lscqp_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    names = declared_functions()
    assert len(names) >= 9
    for n in names:
        assert hasattr(L, n), "liblscqp.so does not export %s" % n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_struct_sizes_match_the_header(api):
    src = '#include <stdio.h>\n#include "lscqp.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(lscqp_header),' \
          ' sizeof(lscqp_row), sizeof(lscqp_box), sizeof(lscqp_info), sizeof(lscqp_class_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    hdr, row, box, info, desc = map(int, out)
    assert (hdr, row, box, info) == (256, 32, 48, 32)  # SURVEY.md §8d byte model
    assert api.HEADER_DTYPE.itemsize == hdr and api.ROW_DTYPE.itemsize == row
    assert api.BOX_DTYPE.itemsize == box and api.INFO_DTYPE.itemsize == info
    assert C.sizeof(api.ClassDesc) == desc


def test_header_field_offsets(api):
    # the shim memcpy's Agent fields into this layout
    d = api.HEADER_DTYPE
    assert [d.fields[f][1] for f in ("p0", "v0", "a0", "goal", "next_waypoint", "vmax", "amax", "radius",
                                     "nominal_velocity", "n_obs", "terminal_segments")] == \
           [0, 24, 48, 72, 96, 120, 144, 168, 176, 184, 188]


def test_create_validates_like_the_reference(api):
    # buildAeqBase throws std::invalid_argument unless n=5, phi=3 (src/traj_optimizer.cpp:198-201)
    with pytest.raises(api.LscqpError) as e:
        api.Solver(api.make_desc(n=4))
    assert e.value.code == api.ERR_INVALID_ARGUMENT and "only n=5, phi=3" in str(e.value)
    with pytest.raises(api.LscqpError) as e:
        api.Solver(api.make_desc(dim=4))  # populatebyrow :249
    assert e.value.code == api.ERR_INVALID_ARGUMENT
    with pytest.raises(api.LscqpError) as e:
        api.Solver(api.make_desc(M=1))
    assert e.value.code == api.ERR_INVALID_ARGUMENT
    s = api.Solver(api.make_desc(M=5, dim=3))
    assert s.nv == 90
    assert s.algorithmic_bytes(20) == 20432
    assert api.Solver(api.make_desc(M=6, dim=3)).algorithmic_bytes(20) == 24464
    assert api.Solver(api.make_desc(M=10, dim=2)).algorithmic_bytes(9) == 18992
    assert s.num_inequalities(20) == 540 + 162 + 138 + 114 + 120
    s.update(api.make_desc(M=5, dim=3, planner_mode=api.PLANNER_DLSC))  # updateParam
    s.close()


def test_no_cpu_fallback(api):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is for GPU-less hosts")
    s = api.Solver(api.make_desc())
    hdr = np.zeros(1, api.HEADER_DTYPE)
    with pytest.raises(api.LscqpError) as e:
        s.solve_host(hdr, None, None, np.zeros(5, api.BOX_DTYPE))
    assert e.value.code == api.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lsc_dr_planner_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.search(r"^\s*(from|import)\s+oracle|lscqp_oracle|orc_solve|oracle/", txt, flags=re.M)
                assert not bad, "%s uses the oracle: %r" % (os.path.join(dp, f), bad.group(0))


def test_next_row_entry_points_validate_and_accept_empty_batches(api):
    """The entry points either side of the QP: empty batches are no-ops, bad arguments are reported (not crashed on), and on
    a GPU-less host every one of them fails loudly instead of computing on the CPU."""
    import torch

    L = api.lib()
    s = api.Solver(api.make_desc(M=5, dim=3))
    h = s._h
    one = (C.c_double * 64)()
    p = C.cast(one, C.c_void_p)
    # empty inputs
    assert L.lscqp_generate_constraints_device(h, api.GEN_CLSC, 0, 8, 0, p, p, p, p, p, p, None) == api.OK
    assert L.lscqp_generate_lsc_device(h, 0, 8, 0, p, p, p, p, p, p, None) == api.OK
    assert L.lscqp_shift_traj_device(h, 0, 1, 1.0, p, p, None) == api.OK
    assert L.lscqp_safety_metrics_device(h, 0, 0, 0, 1, 0.1, 1.0, p, p, p, p, p, None) == api.OK
    assert L.lscqp_validate_step_device(h, 0, 0.2, 1.0, p, p, p, p, p, None) == api.OK
    assert L.lscqp_safety_obstacles_device(h, 0, 0, 0, 1, 0.1, 1.0, p, p, p, 3, p, p, None) == api.OK
    assert L.lscqp_order_by_work_device(0, p, p, None) == api.OK
    assert L.lscqp_launch_capacity(None, 1, 0) == -1 and L.lscqp_launch_capacity(h, -1, 0) == -1
    assert L.lscqp_order_by_cost_device(0, p, p, None) == api.OK and L.lscqp_order_by_cost_device(3, p, None, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_construct_sfc_device_ordered(h, None, api.SFC_INIT, 1, p, p, p, p, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_order_by_work_device(4, None, p, None) == api.ERR_INVALID_ARGUMENT and L.lscqp_order_by_work_device(-1, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_solve_batch_device_ordered(None, 1, 0, p, p, p, p, p, p, p, p, p, 0, p, None) == api.ERR_INVALID_ARGUMENT
    # bad arguments
    assert L.lscqp_safety_obstacles_device(h, 4, 2, 5, 1, 0.1, 1.0, p, p, p, 3, p, p, None) == api.ERR_INVALID_ARGUMENT  # 2 + 4 > 5
    assert L.lscqp_safety_obstacles_device(h, 1, 0, 1, 1, 0.1, 1.0, p, p, p, 3, None, p, None) == api.ERR_INVALID_ARGUMENT  # obstacles announced, no table
    assert L.lscqp_safety_obstacles_device(None, 1, 0, 1, 1, 0.1, 1.0, p, p, p, 0, None, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_generate_constraints_device(h, 7, 4, 8, 0, p, p, p, p, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert b"mode" in L.lscqp_last_error()
    assert L.lscqp_generate_constraints_device(None, api.GEN_LSC, 4, 8, 0, p, p, p, p, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_generate_constraints_device(h, api.GEN_BVC, 4, 8, 0, p, None, p, p, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_safety_metrics_device(h, 4, 2, 5, 1, 0.1, 1.0, p, p, p, p, p, None) == api.ERR_INVALID_ARGUMENT  # 2 + 4 > 5
    assert L.lscqp_shift_traj_device(h, 4, 2, 1.0, p, p, None) == api.ERR_INVALID_ARGUMENT
    assert L.lscqp_construct_sfc(None, api.SFC_INIT, 5, 1, p, p, p, p) == api.ERR_INVALID_ARGUMENT
    hm = C.c_void_p()
    assert L.lscqp_map_create(p, 1, p, p, 0.0, 1.0, C.byref(hm)) == api.ERR_INVALID_ARGUMENT  # resolution 0
    assert L.lscqp_map_create_from_csv(b"/nonexistent/world.csv", p, p, 0.1, 1.0, C.byref(hm)) == api.ERR_INVALID_ARGUMENT
    assert b"cannot open" in L.lscqp_last_error()
    if not torch.cuda.is_available():
        for rc in (L.lscqp_generate_constraints_device(h, api.GEN_CLSC, 4, 8, 0, p, p, p, p, p, p, None),
                   L.lscqp_safety_metrics_device(h, 1, 0, 1, 1, 0.1, 1.0, p, p, p, p, p, None),
                   L.lscqp_safety_obstacles_device(h, 1, 0, 1, 1, 0.1, 1.0, p, p, p, 0, None, p, None),
                   L.lscqp_validate_step_device(h, 1, 0.2, 1.0, p, p, p, p, p, None),
                   L.lscqp_shift_traj_device(h, 1, 1, 1.0, p, p, None)):
            assert rc == api.ERR_NO_DEVICE and b"no CPU fallback" in L.lscqp_last_error()
        wmin, wmax = (C.c_double * 3)(-1, -1, 0), (C.c_double * 3)(1, 1, 1)
        assert L.lscqp_map_create(p, 0, wmin, wmax, 0.1, 1.0, C.byref(hm)) == api.ERR_NO_DEVICE
    s.close()


def test_instance_work_counters_come_with_the_library(api):
    """lscqp_instance_work (no device needed): the work counters of the kernel instance a launch would select, read off the machine
    code when the library was built; every compiled fp64 shape has them and they scale as the kernel does."""
    s5 = api.Solver(api.make_desc(M=5, dim=3))
    small, big = s5.instance_work(64, 20), s5.instance_work(4096, 20)
    assert small["wavefronts"] == 2 and big["wavefronts"] == 1 and "lscqp_pdip_kernel<5,3,true" in small["kernel"]
    for w in (small, big):
        assert w["flops_per_iteration"] > 5 * w["flops_last_pass"] > 0 and w["flops_fixed"] > 0
        assert 2000 < w["f64_insts_per_iteration"] < w["valu_insts_per_iteration"] < 20000
        # flops = 64 lanes x wavefronts x (2 x FMA + other fp64 instructions): between 1x and 2x the instruction count
        lanes = 64 * w["wavefronts"]
        assert lanes * w["f64_insts_per_iteration"] < w["flops_per_iteration"] < 2 * lanes * w["f64_insts_per_iteration"]
    s10 = api.Solver(api.make_desc(M=10, dim=3))
    w10 = s10.instance_work(128, 40)
    assert w10["wavefronts"] == 4 and w10["flops_per_iteration"] > 2 * small["flops_per_iteration"] and w10["lds_bytes"] > 100000
    with pytest.raises(api.LscqpError):
        s5.instance_work(64, 500)  # no instance holds 500 obstacles


def test_instance_table_covers_every_shape_up_to_m10():
    """csrc/lscqp_launch.hpp: every (2 <= M <= 10, dim 2 | 3, with / without the end stop) has a compiled fp64 instance -- the run-time-shaped
    kernel is for M = 11, 12 and for neighbour counts beyond the register slots, not for a planner mode."""
    from lsc_dr_planner_amd import build

    fp64 = {(M, D, E) for (M, D, E, S, W, X) in build.instances() if X == 0}
    missing = [(M, D, E) for M in range(2, 11) for D in (2, 3) for E in (0, 1) if (M, D, E) not in fp64]
    assert not missing, missing
    # ... and holds at least 40 neighbours per agent in registers at the horizons the reference ships (M = 5, 10)
    cap = {}
    for (M, D, E, S, W, X) in build.instances():
        if X == 0:
            cap[(M, D, E)] = max(cap.get((M, D, E), 0), S * max(1, 64 * W // (6 * M - 3)))
    for key in ((5, 3, 1), (5, 3, 0), (10, 2, 1), (10, 2, 0), (10, 3, 1)):
        assert cap[key] >= 40, (key, cap[key])


def test_python_constants_are_the_headers(api):
    """Statuses, return codes and lscqp_info flags used by the tests and bench.py (api.py) are include/lscqp.h's, name for name."""
    import os
    import re

    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "lscqp.h")).read()
    hdr = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(LSCQP_(?:STATUS|ERR|INFO)_[A-Z0-9_]+|LSCQP_OK)\s*=?\s+(\d+)\b", txt)}
    hdr.update({m.group(1): int(m.group(2)) for m in re.finditer(r"\b(LSCQP_(?:STATUS|ERR)_[A-Z0-9_]+|LSCQP_OK) = (\d+)", txt)})
    seen = 0
    for name, val in vars(api).items():
        if re.fullmatch(r"(STATUS|ERR|INFO)_[A-Z0-9_]+|OK", name) and isinstance(val, int):
            assert hdr.get("LSCQP_" + name) == val, (name, val, hdr.get("LSCQP_" + name))
            seen += 1
    assert seen >= 16 and hdr["LSCQP_INFO_RESCUED"] == 32
