"""Assembly-kernel micro-benchmark on the bench mesh: average launch time (HIP events) of the fused residual+Jacobian
kernel and of the tau-moment kernel, plus checksums of F and of the CSR values (variants must agree bit for bit).
usage: asm_bench.py [variant-name ...]   ('' or 'base' = the in-tree libcfdh.so); each variant runs in a child process."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(variant, m):
    from cfd_hemodynamic_amd import _lib
    if variant not in ("", "base"):
        _lib._SO = os.path.join(ROOT, "cfd_hemodynamic_amd", "variants", "libcfdh_%s.so" % variant)
    import numpy as np
    if os.environ.get("ASM_BENCH_3D"):
        from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
        sc = MicrovasculatureSimulation("stabilized_schur", 0.01, 1.0, res=float(os.environ["ASM_BENCH_3D"]), quiet=True,
                                        options=dict(remove_p_mean=0))
    else:
        from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
        sc = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=m, quiet=True)
    ctx = sc.solver.ctx
    sc.solver.solveStep(); sc.solver.advance(); sc.solver.solveStep()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(40):
        ctx.assemble(True)
    t, n = ctx.profile_get(0)
    e, ne = ctx.profile_get(7)
    F = np.concatenate(ctx.get_residual())
    J = ctx.get_csr()
    h = hashlib.sha1(F.tobytes() + J.data.tobytes()).hexdigest()[:12] + " |F| %.15e |J| %.15e" % (np.linalg.norm(F), np.linalg.norm(J.data))
    print("%-12s asm %.1f us (%d launches; empty event pair %.1f us)  blocks %d  sha %s" % (
        variant or "base", 1e3 * t / n, n, 1e3 * e / max(ne, 1), ctx.info(7), h), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run(sys.argv[2], int(sys.argv[3]))
    else:
        m = int(os.environ.get("ASM_BENCH_M", "200"))
        for v in (sys.argv[1:] or ["base"]):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", v, str(m)])
