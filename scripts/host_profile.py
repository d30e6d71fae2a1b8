"""cProfile of the host side of one training step at 64^3 (is the step launch-bound, and where?)."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "direct64"]
model, crit, opt = bench.build(wl, dev)
params = list(model.parameters())
xr, ct = bench.make_batch(wl, 0, dev)
for _ in range(3): bench.train_step(model, params, crit, opt, xr, ct)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): bench.train_step(model, params, crit, opt, xr, ct)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {1e3*t_issue/5:.2f} ms/step, wall {1e3*t_all/5:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): bench.train_step(model, params, crit, opt, xr, ct)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
