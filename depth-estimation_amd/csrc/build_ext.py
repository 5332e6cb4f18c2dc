"""Builds lib/lattice_ext.so: the pybind torch extension `filter(src, ref)` over the C ABI
(csrc/lattice_ext.cpp).  Plain g++ against the installed torch headers; no device code."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "lib")
OUT = os.path.join(LIB, "lattice_ext.so")
SRC = os.path.join(HERE, "lattice_ext.cpp")


def build(force=False):
    deps = [SRC, os.path.join(HERE, "..", "..", "include", "phl.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce

    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include",
                                os.path.abspath(os.path.join(HERE, "..", "..", "include"))]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=lattice_ext", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w"]
    cmd += [f"-I{i}" for i in inc]
    cmd += [SRC, "-o", OUT, f"-L{tl}", "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-lc10_hip", f"-L{LIB}", "-lphl",
            f"-Wl,-rpath,{tl}", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
