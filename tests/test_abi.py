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
