#!/usr/bin/env python3
"""Device-built vs host-built AMG hierarchies: iteration counts and set-up time.
  python tools/amg_dev_check.py c3 200 6     (config, size parameter, steps); env CFDH_AMG_HOST=1 / CFDH_AMG_AGG=host select variants."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("CFDH_IMPORT_TORCH_FIRST") == "1":
    import torch  # noqa: F401  (the process then runs on the HIP runtime torch bundles, like bench.py)
    torch.cuda.set_device(0)
import bench
cfg, size, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
args = types.SimpleNamespace(config=cfg, m=200, nx=288, ny=115, res=7.3e-6, res3=2e-4, dt=0.001 if cfg == "c5" else 0.01, ramp=0.03,
                             v_max={"c5": 0.05, "c5b": 1.5}.get(cfg, 100.0))
setattr(args, {"c3": "m", "c2": "nx", "c4": "ny", "c5": "res", "c5b": "res3"}[cfg], float(size) if cfg in ("c5", "c5b") else int(size))
verbose = int(os.environ.get("VERBOSE", "0"))
t0 = time.perf_counter()
sc = bench.make_scenario(args, "stabilized_schur", device=0, verbose=verbose)
t_setup = time.perf_counter() - t0
s = sc.solver
if os.environ.get("CFDH_DUMP_MAPS"):  # which libraries are mapped where (to resolve the frames of a native backtrace)
    open(os.environ["CFDH_DUMP_MAPS"], "w").write(open("/proc/self/maps").read())
its, pcs, walls = [], [], []
for k in range(steps):
    bench.step_hook(sc, k, args.dt)
    t0 = time.perf_counter()
    s.solveStep(); s.advance()
    walls.append(time.perf_counter() - t0)
    its.append(s.last_stats.krylov_its); pcs.append(s.last_stats.ms_pc_setup)
print("variant host=%s agg=%s | %s size %s nv=%d | setup %.2fs | its %s | pc_setup ms %s | step wall ms %s" % (
    os.environ.get("CFDH_AMG_HOST", "0"), os.environ.get("CFDH_AMG_AGG", "dev"), cfg, size, sc.mesh.num_vertices, t_setup, its,
    ["%.1f" % p for p in pcs], ["%.1f" % (1e3 * w) for w in walls]))
print("  L2 norms: %.12e %.12e | guessed solves %d, mean |r0|/|b| %.2e | mean step wall of the last half %.2f ms" % (s.functional(2), s.functional(3), s.ctx.info(70), 1e-6 * s.ctx.info(71), 1e3 * sum(walls[len(walls) // 2:]) / max(1, len(walls) - len(walls) // 2)))
