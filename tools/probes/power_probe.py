"""Power / clock of a sustained GEMM loop: ours vs hipBLASLt (torch.matmul), sampled with rocm-smi in a thread."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
M, N, K = 26112, 7680, 2560
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
samples = []
stop = False
def poll():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True)
        samples.append((time.time(), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[:200]))
        time.sleep(0.3)
def run(name, fn, secs=4.0):
    global stop, samples
    samples, stop = [], False
    th = threading.Thread(target=poll); th.start()
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    stop = True; th.join()
    print(f"== {name}: {2.0*M*N*K*n/dt/1e12:.0f} TF sustained")
    for t, s in samples[2:8]: print("   ", s[:300])
r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True)
print(r.stdout[:600])
run("ours v9", lambda: ops.gemm(x, w, variant=9))
run("hipBLASLt", lambda: torch.matmul(x, w.t()))
run("ours v3", lambda: ops.gemm(x, w, variant=3))
