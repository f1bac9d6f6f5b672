"""Build the C++ shim's test driver (plain g++: the shim only talks to the C ABI of liblscqp.so)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(HERE, "shim_test")


def build(force=False):
    srcs = [os.path.join(HERE, "src", "traj_optimizer.cpp"), os.path.join(HERE, "src", "goal_optimizer.cpp"),
            os.path.join(HERE, "test", "shim_test.cpp")]
    deps = srcs + [os.path.join(HERE, "include", f) for f in os.listdir(os.path.join(HERE, "include"))] + [
        os.path.join(ROOT, "include", "lscqp.h")]
    lib = os.path.join(os.path.dirname(HERE), "liblscqp.so")
    if not force and os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(d) for d in deps + [lib]):
        return EXE
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include")] + srcs + [
        "-L", os.path.dirname(HERE), "-llscqp", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


if __name__ == "__main__":
    print(build(force=True))
