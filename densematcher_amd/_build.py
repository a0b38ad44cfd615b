"""
Build libdensematch.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

    python -m densematcher_amd._build            # incremental
    python -m densematcher_amd._build --force

hipcc cross-compiles without a GPU; the .so is built IN-TREE next to this file so
that it travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdensematch.so")
SOURCES = ["dm_ctx.hip", "dm_p2p.hip", "dm_fmap.hip", "dm_project.hip", "dm_zoomout.hip", "dm_zoomfuse.hip", "dm_icp.hip", "dm_simnn.hip", "dm_knnsplit.hip", "dm_energy.hip", "dm_assign.hip", "dm_precise.hip", "dm_eigen.hip", "dm_peaks.hip", "dm_lbfgs.hip", "dm_fitfuse.hip", "dm_laplacian.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "densematch.h"),
                                                                 os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


LIB_EXP = os.path.join(HERE, "libdensematch_exp.so")


def build_experiments(force=False, verbose=False):
    """The same sources with -DDM_EXPERIMENTS: ablation knobs read from the environment (DM_SIMNN_DEBUG, DM_GRED_DEBUG,
    DM_SOLVE_DEBUG ...), some of which give WRONG results.  Used by tools/*_experiment.py only; a separate file that
    nothing in the package loads."""
    return build(force=force, verbose=verbose, extra_flags=("-DDM_EXPERIMENTS",), lib=LIB_EXP, objdir=os.path.join(BUILD, "exp"))


def build(force=False, verbose=False, extra_flags=(), lib=None, objdir=None):
    LIB = lib or globals()["LIB"]
    BUILD = objdir or globals()["BUILD"]
    os.makedirs(BUILD, exist_ok=True)
    newest = _newest_dep()
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose=True))
