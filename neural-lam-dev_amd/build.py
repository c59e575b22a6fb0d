"""Builds libnlam_hip.so in-tree with hipcc for gfx950 (no cmake, no libtorch)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnlam_hip.so")
FLAGS = ["--offload-arch=gfx950", "-mcode-object-version=5", "-O3", "-std=c++17", "-fPIC"]
# extra flags for experiments, e.g. NLAM_HIPCC_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp"
FLAGS += os.environ.get("NLAM_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _file_flags(src):
    """Extra hipcc flags a source asks for in a `// NLAM_HIPCC_FLAGS: ...` line of its header."""
    with open(src) as f:
        for line in f.readlines()[:60]:
            if line.startswith("// NLAM_HIPCC_FLAGS:"):
                return line.split(":", 1)[1].split()
    return []


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(HERE, "..", "include", "*.h")
    )
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + _file_flags(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc] + FLAGS + ["-shared", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
