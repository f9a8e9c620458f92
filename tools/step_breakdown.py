"""Where a step of a bench config (c3 | c2 | c4 | c5 | c5b) spends its wall-clock time: inlet update (step_hook), solveStep, wall shear stress, state copy."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
args = types.SimpleNamespace(config=cfg, m=200, nx=288, ny=115, res=7.3e-6, res3=2e-4, dt=0.001 if cfg == "c5" else 0.01, ramp=0.03,
                             v_max={"c5": 0.05, "c5b": 1.5}.get(cfg, 100.0))
sc = bench.make_scenario(args, "stabilized_schur", device=0)
s = sc.solver
s.initStressForm()
acc = {"hook": 0.0, "solve": 0.0, "solve_lib_ms_total": 0.0, "wss": 0.0, "advance": 0.0}
n0, n1 = (36, 56) if cfg == "c5" else (6, 26)
for k in range(n1):
    t0 = time.perf_counter(); bench.step_hook(sc, k, args.dt); torch.cuda.synchronize(); t1 = time.perf_counter()
    s.solveStep(); torch.cuda.synchronize(); t2 = time.perf_counter()
    s.assemble_wss(); torch.cuda.synchronize(); t3 = time.perf_counter()
    s.advance(); torch.cuda.synchronize(); t4 = time.perf_counter()
    if k >= n0:
        acc["hook"] += t1 - t0; acc["solve"] += t2 - t1; acc["wss"] += t3 - t2; acc["advance"] += t4 - t3
        acc["solve_lib_ms_total"] += 1e-3 * s.last_stats.ms_total
print("nv", sc.mesh.num_vertices, {k: "%.2f ms" % (1e3 * v / (n1 - n0)) for k, v in acc.items()})
