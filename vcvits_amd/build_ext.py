"""Build libvcvits_hip.so (hipcc, gfx950) in-tree.

`python -m vcvits_amd.build_ext` or `__graft_entry__.build()`.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvcvits_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _deps(obj, src, hdrs):
    """Headers a source really includes: from the compiler's depfile (`-MMD`) when the last build left one, every header
    of csrc/ otherwise."""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return [src] + hdrs
    words = open(d).read().replace("\\\n", " ").split()
    inc = [os.path.normpath(os.path.join(CSRC, w)) for w in words[1:] if not w.endswith(":")]
    return [src] + [w for w in inc if w.startswith(ROOT)]


def build(force=False, verbose=True):
    srcs = _sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "vcvits_hip.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, _deps(o, s, hdrs)):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-MMD", "-MF", o[:-2] + ".d", "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
