"""Build libaniportrait_hip.so (gfx950) in-tree with hipcc.  `python -m aniportrait_amd.build`.

The shared object lands in aniportrait_amd/lib/ (git-ignored; it travels to the GPU box with the
gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libaniportrait_hip.so")
SOURCES = ["gemm.hip", "gemm2.hip", "ffn.hip", "tblock.hip", "norm.hip", "convbn.hip", "attention.hip", "attn_dma.hip", "elementwise.hip", "api.cpp"]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "aniportrait_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=(), lib=None):
    """extra_flags / lib: experiment builds (e.g. -DANIP_GELU_EXACT into lib/libaniportrait_hip_exact.so, loaded with
    ANIP_LIB=<path>); the product is always the flag-less default."""
    global LIB
    if lib is not None:
        LIB, force = lib, True
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + (".x.o" if extra_flags else ".o"))
        cmd = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *extra_flags, "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    out = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")), None)
    print(build(force="--force" in sys.argv, extra_flags=flags, lib=os.path.join(LIBDIR, out) if out else None))
