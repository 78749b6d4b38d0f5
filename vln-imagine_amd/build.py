"""Builds libvlni.so (every HIP kernel + the C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvlni.so")
SOURCES = ["api.hip", "gemm.hip", "layernorm.hip", "elementwise.hip", "attention.hip", "graphmap.hip", "blocks.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source flags. attention.hip: MFMA results in VGPRs - with the accumulators in AGPRs every softmax operation on a score tile is an
# accvgpr read + write around it (448 of the forward kernel's 1872 VALU instructions, 144 registers instead of 113: -16 % per launch)
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
if os.environ.get("VLNI_DIAG") == "1":       # stamped / timing-only kernel builds + vlni_debug_pk_stamps (tools/gemm_stamps.py); never the default
    FLAGS.append("-DVLNI_DIAG")


LAST = {}        # what the last build() call did: {"compiled": [sources], "linked": bool, "reused": bool}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    force = force or os.environ.get("VLNI_FORCE_BUILD") == "1"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]     # *_impl.inc: kernel bodies included per 16-bit type
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs + [os.path.abspath(__file__)]):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    linked = False
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        linked = True
    # one line that tells a build from a no-op (`build_exercised` was not observable)
    LAST.update(compiled=[os.path.basename(s) for s, _ in jobs], linked=linked, reused=not jobs and not linked)
    if verbose:
        sys.stderr.write(f"[vlni build] compiled {len(jobs)}/{len(SOURCES)} sources for gfx950 ({', '.join(LAST['compiled']) or 'none'}), "
                         f"{'linked' if linked else 'kept'} {os.path.relpath(LIB)}" + (" (up to date: reused)" if LAST['reused'] else "") + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
