#!/usr/bin/env python
"""Build a variant of the library for in-box A/B runs: one source recompiled with extra -D flags, the other objects reused.
  python tools/build_variant.py <tag> <source relative to csrc, e.g. kernels/conv3x3_wino2d_f32.hip> -DW2_ABL=1 ...
-> gddim_amd/libgddim_<tag>.so   (load with GDDIM_LIB=gddim_amd/libgddim_<tag>.so)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gddim_amd import build as b  # noqa: E402


def main():
    tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build()
    hipcc = b._hipcc()
    objs = []
    for s in b.SOURCES:
        obj = os.path.join(b.BUILD, s.replace("/", "_") + ".o")
        if s == src:
            obj = os.path.join(b.BUILD, s.replace("/", "_") + "." + tag + ".o")
            cmd = [hipcc, "--offload-arch=" + b.ARCH, "-O3", "-std=c++17", "-fPIC"] + flags + ["-c", os.path.join(b.CSRC, s), "-o", obj]
            if s.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(b.HERE, "libgddim_%s.so" % tag)
    subprocess.check_call([hipcc, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-Wl,-z,defs", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
