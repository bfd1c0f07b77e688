"""TEST INFRASTRUCTURE: build tests/_ref/libsml_hip_prepref.so -- the product's sources compiled with
-DSML_TEST_PREP_REFERENCE, which textually includes tests/csrc/prep_cub_reference.inc / prep_cub_kernels.inc: the hipCUB
radix-sort path of the index preparation that index_prep.hip replaced.  Only the A/B tests load it (SML_PREP=cub through a
HipEngine(lib=...)): the product library has no library sort and no hipcub / rocprim symbol.

    python tests/build_reference.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libsml_hip_prepref.so")


def build(force=False, verbose=False):
    sys.path.insert(0, REPO)
    from sml_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(B.CSRC, h) for h in B.HEADERS] + [os.path.join(HERE, "csrc", f) for f in ("prep_cub_reference.inc", "prep_cub_kernels.inc")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for s in B.SOURCES:
        src = os.path.join(B.CSRC, s)
        obj = os.path.join(OUT, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or B._stale(obj, [src] + deps):
            cmd = [hipcc] + B.FLAGS + ["-DSML_TEST_PREP_REFERENCE", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if out and (verbose or p.returncode != 0):
            sys.stderr.write(out.decode(errors="replace"))
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or B._stale(LIB, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
