"""Build liblra_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblra_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(f) > t for f in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if (not force and os.path.exists(o) and os.path.getmtime(o) > os.path.getmtime(s)
                and all(os.path.getmtime(o) > os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")) +
                        glob.glob(os.path.join(HERE, "..", "include", "*.h")))):
            continue
        cmd = ["hipcc", *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
        if verbose and out:
            print(out.decode())
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=False, verbose=True))
