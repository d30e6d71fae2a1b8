"""Dev aid: LayerNorm forward / backward kernel timing at the block shapes."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
from hvc import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
from hvc import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rows, C, rpb in ((65536, 256, 32768), (16384, 256, 4096)):
    x = torch.randn(rows, C, device=dev); g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    sc = torch.randn(rows // rpb, C, device=dev) * 0.1; sh = torch.randn(rows // rpb, C, device=dev) * 0.1
    dy = torch.randn(rows, C, device=dev, dtype=torch.bfloat16); dres = torch.randn(rows, C, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, sc, sh, rows_per_batch=rpb, out_dtype=torch.bfloat16)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b, sc, sh, rows_per_batch=rpb, out_dtype=torch.bfloat16))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, b, sc, mean, rstd, dres=dres, rows_per_batch=rpb))
    print(f"{os.path.basename(_lib.LIB_PATH)} LN {rows}x{C}: fwd {tf*1e6:.1f} us  bwd {tb*1e6:.1f} us (incl. final reduction + allocs)", flush=True)
