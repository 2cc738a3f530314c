"""Build libcogview_hip.so (gfx950) from the .hip sources in this directory with hipcc.

No torch involvement: the library is a plain C-ABI shared object (see include/cogview_hip.h).
Objects go to build/ (git-ignored); the .so is written in-tree to cogview_amd/lib/ so that it travels to
the GPU box with the repo snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcogview_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{HERE}"]
# A/B builds (tools/evidence.sh documents the A/B workflow): COGV_HIPCC_EXTRA="-DX -DY" adds flags, COGV_VARIANT=name redirects the objects to
# build/obj_<name>/ and the library to build/ab/libcogview_<name>.so (select it at run time with COGVIEW_HIP_LIB)
FLAGS += os.environ.get("COGV_HIPCC_EXTRA", "").split()
if os.environ.get("COGV_VARIANT"):
    OBJ_DIR = os.path.join(ROOT, "build", "obj_" + os.environ["COGV_VARIANT"])
    LIB_DIR = os.path.join(ROOT, "build", "ab")
    LIB_PATH = os.path.join(LIB_DIR, "libcogview_%s.so" % os.environ["COGV_VARIANT"])


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libcogview_hip.so")


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


# Extra translation units compiled from a source that is already in sources(), with a macro selecting what is
# instantiated: the generation-4 GEMM kernel's 2 dtypes x 4 layouts build as eight objects in parallel
# (gemm.hip -DCOGV_W4_TU=k defines only cogv_w4_launch_k).
EXTRA_UNITS = [("gemm.hip", f"gemm_w4_{k}.o", [f"-DCOGV_W4_TU={k}"]) for k in range(8)]
# the skinny-M kernels of the decode step, one unit per dtype (gemv.hip is empty without the macro)
EXTRA_UNITS += [("gemv.hip", f"gemv_{k}.o", [f"-DCOGV_GEMV_TU={k}"]) for k in range(2)]


def units():
    """(source, object, extra flags) of every object that goes into the library."""
    out = [(src, os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o"), []) for src in sources()]
    out += [(os.path.join(HERE, src), os.path.join(OBJ_DIR, obj), flags) for src, obj, flags in EXTRA_UNITS]
    return out


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cuh", ".h"))]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hdrs)


def scan_asm_hazards(lines):
    """Static check of a unit's device assembly for the one thing the compiler cannot know about the hand-issued
    asynchronous operations of these kernels: loads issued inside inline asm (global_load_* into registers, ds_read_*)
    whose result is awaited by a LATER `s_waitcnt` (asm, counted).  Returns [(kernel, line, text)] for every instruction
    that reads a destination register of such a load while it may still be in flight -- e.g. a register copy the
    compiler placed in front of the wait at a control-flow join, or an epilogue that reuses the registers before a bare
    wait: both silently compute on stale data (the first shipped for an hour in round 2 and only failed when other tests
    had warmed the clocks).  Linear scan in layout order per kernel: asm loads are tracked oldest first per counter
    (vmcnt / lgkmcnt); `s_waitcnt <cnt>(N)` retires all but the N youngest; labels do not reset the state.
    Timing-probe instantiations (COGV_CONV_EXP: wrong results by design) are skipped."""
    import re

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    out, func, in_asm = [], None, False
    flight = {"vmcnt": [], "lgkmcnt": []}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            func, flight = m.group(1), {"vmcnt": [], "lgkmcnt": []}
            if re.search(r"conv_kernelILb[01]ELi[1-9]\d*EEEv", func):
                func = None
            continue
        if func is None:
            continue
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        toks = [x.strip() for x in rest.split(",")]
        if in_asm and op.startswith("global_load") and not op.startswith("global_load_lds"):
            flight["vmcnt"].append(regs(toks[0]))
            continue
        if in_asm and op.startswith("global_atomic") and " sc0" in t:        # returning atomic: toks[0] is the destination
            flight["vmcnt"].append(regs(toks[0]))
            continue
        if in_asm and op.startswith("ds_read"):
            flight["lgkmcnt"].append(regs(toks[0]))
            continue
        if in_asm and op.startswith("ds_write"):       # no destination, but it takes a slot of the in-order LDS counter
            flight["lgkmcnt"].append(set())
            continue
        if op.startswith("s_waitcnt"):
            for cnt in ("vmcnt", "lgkmcnt"):
                m = re.search(cnt + r"\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    flight[cnt] = flight[cnt][len(flight[cnt]) - n:] if 0 < n < len(flight[cnt]) else ([] if n == 0 else flight[cnt])
            continue
        pending = set()
        for fl in flight.values():
            for r in fl:
                pending |= r
        if pending:
            srcs = set()
            for tok in (toks if op.startswith(("ds_write", "global_store", "buffer_store", "global_atomic")) else toks[1:]):
                srcs |= regs(tok)
            # a scratch access while asm loads are in flight is the compiler spilling around them (the k-loops keep
            # asm reads in flight all the time): refuse it whatever registers it names.  Spills and scratch arrays
            # elsewhere (the generation-3 GEMM parks values across its item loop, the generic epilogue keeps a small
            # array) cannot sit between an asm load and its wait and are tolerated.
            if srcs & pending or op.startswith("scratch_"):
                out.append((func, i + 1, t))
    return out


def asm_load_hazards(src, extra=()):
    """scan_asm_hazards() of one source compiled on its own (tools / tests)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([_hipcc()] + FLAGS + list(extra) + ["--cuda-device-only", "-S", src, "-o", asm], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr}")
        return scan_asm_hazards(open(asm).read().splitlines())


def _compile(src, obj, extra=()):
    import glob, tempfile
    # -save-temps=obj leaves the device assembly next to the object (no second compile): it feeds scan_asm_hazards
    with tempfile.TemporaryDirectory(dir=os.path.dirname(obj)) as td:
        tobj = os.path.join(td, os.path.basename(obj))
        cmd = [_hipcc()] + FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-save-temps=obj", "-c", src, "-o", tobj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        hz = []
        for asm in glob.glob(os.path.join(td, "*amdgcn*.s")):
            hz += scan_asm_hazards(open(asm).read().splitlines())
        if hz:
            raise RuntimeError(f"{src}: a register of an in-flight asm load is read before its s_waitcnt:\n" +
                               "\n".join(f"  {k} line {n}: {t}" for k, n, t in hz[:10]))
        os.replace(tobj, obj)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    dep_t = _deps_mtime()
    jobs, objs = [], []
    for src, obj, flags in units():
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_t):
            jobs.append((src, obj, flags))
    if jobs:
        if verbose:
            print(f"[cogview_amd] hipcc {ARCH}: compiling {len(jobs)} file(s)", flush=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(jobs))) as ex:
            list(ex.map(lambda a: _compile(*a), jobs))
    if jobs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[cogview_amd] linked {LIB_PATH}", flush=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
