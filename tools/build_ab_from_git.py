"""A/B baseline: build a library whose gemm.hip (9 objects) comes from a git revision, every other object from the current
build.  python tools/build_ab_from_git.py <rev> <tag>  ->  build/ab/libcogview_<tag>.so (select with COGVIEW_HIP_LIB)."""
import concurrent.futures, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cogview_amd.csrc import build as B

rev, tag = sys.argv[1], sys.argv[2]
files = sys.argv[3:] or ["gemm.hip"]
out_dir = os.path.join(ROOT, "build", "ab")
os.makedirs(out_dir, exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    for f in os.listdir(B.HERE):
        if f.endswith((".cuh", ".h", ".hip")):
            data = subprocess.run(["git", "show", f"{rev}:cogview_amd/csrc/{f}"], cwd=ROOT, capture_output=True).stdout \
                if f in files or f.endswith((".cuh", ".h")) else open(os.path.join(B.HERE, f), "rb").read()
            open(os.path.join(td, f), "wb").write(data)
    flags = [x if not x.startswith("-I" + B.HERE) else "-I" + td for x in B.FLAGS]
    jobs, objs = [], []
    for src, obj, extra in B.units():
        name = os.path.basename(src)
        if name in files:
            o2 = os.path.join(td, os.path.basename(obj))
            jobs.append([B._hipcc()] + flags + extra + ["-c", os.path.join(td, name), "-o", o2])
            objs.append(o2)
        else:
            objs.append(obj)
    with concurrent.futures.ThreadPoolExecutor(max_workers=9) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            if r.returncode:
                raise SystemExit(r.stderr[-3000:])
    lib = os.path.join(out_dir, f"libcogview_{tag}.so")
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib] + objs, check=True)
    print("built", lib)
