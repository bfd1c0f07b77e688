"""Build libsml_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsml_hip.so")
SOURCES = ["transfer_net.hip", "mf_kernels.hip", "index_prep.hip", "capi.hip"]
HEADERS = ["sml_dev.h", "sml_kernels.h", "transfer_fwd_body.inc", os.path.join("..", "..", "include", "sml_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # leading scalar / pointer kernel parameters arrive in SGPRs with the wavefront (up to 16 SGPRs) instead of behind a
         # scalar-load round trip of the argument segment: 0.22-0.24 us per dependent kernel (tools/launch_boundary_probe.hip)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
FLAGS += os.environ.get("SML_EXTRA_FLAGS", "").split()      # measurement builds, e.g. -DSML_TIMELINE (tools/timeline_probe.py)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if out and (verbose or p.returncode != 0):
            sys.stderr.write(out.decode(errors="replace"))
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
