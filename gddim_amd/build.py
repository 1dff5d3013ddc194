"""Build the native library (HIP kernels for gfx950 + C ABI) in-tree: gddim_amd/libgddim_amd.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs anywhere the ROCm toolchain is present.
Objects are cached under gddim_amd/_build and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libgddim_amd.so")
SOURCES = ["engine.cpp", "plan.cpp", "options.cpp", "host_math.cpp", "kernels/igemm.hip", "kernels/igemm_bf16x3.hip", "kernels/conv3x3_halo_bf16x3.hip", "kernels/conv3x3_wino2d_f32.hip", "kernels/conv3x3_wino42_f32.hip", "kernels/conv3x3_wino2d_split6.hip", "kernels/attn_fused_f32.hip", "kernels/norm_act.hip", "kernels/update.hip"]
HEADERS = ["common.h", "plan.h", "options.h", "kernels/qstats.h", "kernels/epilogue.h", "kernels/rng.h", "kernels/act.h", "../../include/gddim_amd.h"]
ARCH = "gfx950"
# per-file flags: the split6 kernel wants its fp32 adds scalar (v_pk_add_f32 costs issue time beside bf16 MFMAs)
EXTRA_FLAGS = {"kernels/conv3x3_wino2d_split6.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    hdr_mtime = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(BUILD, src.replace("/", "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_mtime):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-z,defs", "-o", LIB] + objs  # undefined symbols fail the link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
