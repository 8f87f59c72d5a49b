"""Builds libmpopis_hip.so (gfx950) in-tree with hipcc.  `python -m mpopis_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmpopis_hip.so")
SOURCES = ["engine_api.hip", "engine_ais.hip", "engine_harness.hip", "engine_comm.hip", "kernels_rollout.hip", "kernels_reweight.hip",
           "kernels_sample.hip", "kernels_linalg.hip", "kernels_select.hip", "kernels_ce.hip", "kernels_cma.hip", "kernels_invsqrt.hip", "kernels_mfma.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _newer(a, deps):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mpopis.h"))
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and _newer(obj, [sp] + headers):
            continue
        cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
