"""Dev aid: timing-only knock-out builds of the GEMM kernel (build_ab/lib_gko<N>.so = gemm.hip compiled with -DHVC_GEMM_KO=N, wrong
results by design) against the shipped library on the block's token-matrix shapes.  1 no epilogue, 2 no global operand loads,
3 no MFMA, 4 no LDS commit, 5 no LDS fragment reads."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "hybrid-vit-cascade_amd"))
from hvc import _lib
_lib.LIB_PATH = sys.argv[1]
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M = 65536
out = []
for (N, K) in ((256, 256), (768, 256), (1024, 256), (256, 1024)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    out.append(f"N{N} K{K} {timeit(lambda: ops.gemm(x, w)):6.1f}")
print("  ".join(out))
''' % ROOT
libs = [os.path.join(ROOT, "hybrid-vit-cascade_amd/lib/libhvc_hip.so")] + [os.path.join(ROOT, f"build_ab/lib_gko{k}.so") for k in (1, 2, 3, 4, 5)]
for pers in ("0", "1"):
    for lib in libs:
        env = dict(os.environ, HVC_GEMM_PERSISTENT=pers)
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, env=env)
        print(f"persistent={pers} {os.path.basename(lib):16s} | {r.stdout.strip()}", flush=True)
        if r.returncode: print(r.stderr[-1500:])
