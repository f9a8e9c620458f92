#!/usr/bin/env python3
"""Headline benchmark: time steps per second of the stabilized_schur step.

  python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5]

Default workload = BASELINE.json configs[2], the one its metric is quoted on: DFG 2D-1 refined to ~1M P1/P1 DOF.
Other configs (same harness, for the per-config lines in DESIGN.md): c2 lid-driven cavity nx=288 (250 k DOF),
c4 stenosis "moderate" at the reference geometry (~2 M DOF), c5 stenosis with vascular tree, pulsatile inlet,
dt = 0.001 (~8 M DOF).

N=1 runs in-process; for N>1 the driver launches one rank per GPU with torch.distributed.run.  A "step" is one full
time step (moments, Newton with fused residual/Jacobian assembly, FGMRES + Schur/AMG preconditioner, wall shear
stress, u_prev update) with every field resident in HBM.  Rank 0 prints ONE JSON line carrying `roofline` (dominant
instrumented kernel, HIP-event timed), and at N=1 `cpu_baseline` (the CPU oracle running the same algorithm on the
same mesh for a bounded number of steps on the host cores) and `parity` (GPU vs oracle functionals with both sides
converged tightly).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def kernel_source_sha16():
    """Fingerprint of the kernel sources (csrc/*.hip, *.cpp, *.hpp, include/*.h incl. the quadrature tables): ties a PMC summary to the code it measured
    (the GPU box has no .git, so a commit id is not available at run time)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cfd_hemodynamic_amd", "csrc")
    inc = os.path.join(ROOT, "include")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(inc, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel_bytes(ctx):
    """Algorithmic HBM bytes per launch of the instrumented kernels (DESIGN.md section 'Kernels')."""
    nvo, nnzv, nc, spnnz = (ctx.info(k) for k in (0, 3, 2, 4))
    if ctx.info(26) == 3 and ctx.info(28) in (1, 2):
        # hexahedra / P2 tetrahedra (csrc/cfdh_gen3.hip): block values out (128 B per graph entry), per node coordinates, iterate, u_prev,
        # Dirichlet data in (~110 B) and the residual out (32 B), connectivity per cell; the staged blocks (16 + 4 doubles per lane,
        # written and read once) are part of the algorithm
        nl = ctx.info(29)
        return {0: ("gen3_asm_residual_jacobian", 128.0 * nnzv + 142.0 * nvo + (4.0 * nl + 2.0 * 160.0 * nl * nl) * nc),
                1: ("spmv3_full_block4x4", 132.0 * nnzv + 68.0 * nvo)}
    if ctx.info(26) == 3:
        # tetrahedra: 4x4 vertex blocks (128 B of values + 4 B column per graph entry), 4 dofs per vertex
        kb = {
            0: ("asm3_residual_jacobian", 128.0 * nnzv + 32.0 * nvo + (16.0 + 96.0) * nc + 80.0 * nvo),
            1: ("spmv3_full_block4x4", 132.0 * nnzv + 68.0 * nvo),
            2: ("tau_moments_tet", 16.0 * nc + 48.0 * nvo + 96.0 * nc),
        }
        if ctx.info(25):
            kb[4] = ("amg_up0_pressure", 8.0 * ctx.info(19) + 16.0 * nvo + 8.0 * ctx.info(23))
            kb[5] = ("amg_up0_velocity_3rhs", 8.0 * ctx.info(20) + 48.0 * nvo + 24.0 * ctx.info(24))
            kb[8] = ("amg_down0_pressure", 8.0 * ctx.info(21) + 12.0 * ctx.info(23) + 8.0 * nvo)
            kb[9] = ("amg_down0_velocity_3rhs", 8.0 * ctx.info(22) + 28.0 * ctx.info(24) + 24.0 * nvo)
        return kb
    if ctx.info(28) in (1, 2):
        # P2 / Q1 nodal elements (csrc/cfdh_gen.hip): block values out (72 B per graph entry), per node coordinates, iterate,
        # u_prev, Dirichlet data in (~80 B) and the residual out (24 B), connectivity and value slots per cell
        nl = ctx.info(29)
        kb = {0: ("gen_asm_residual_jacobian", 72.0 * nnzv + 104.0 * nvo + 4.0 * (nl + nl * nl) * nc),
              1: ("spmv_full_block3x3", 76.0 * nnzv + 52.0 * nvo)}
        if ctx.info(25):
            kb[4] = ("amg_up0_pressure", 8.0 * ctx.info(19) + 16.0 * nvo + 8.0 * ctx.info(23))
            kb[5] = ("amg_up0_velocity_2rhs", 8.0 * ctx.info(20) + 32.0 * nvo + 16.0 * ctx.info(24))
        return kb
    kb = {
        # fused residual+Jacobian: SURVEY.md 8d figure, 624 B per vertex
        0: ("asm_residual_jacobian", 624.0 * nvo),
        # block SpMV: values 72 B + column 4 B per graph entry; rowptr 4, x 24, y 24 per row
        1: ("spmv_full_block3x3", 76.0 * nnzv + 52.0 * nvo),
        # tau moments: cells 12 + coords/u_prev gathers (L2-resident per vertex: 32 B) + 64 B record out
        2: ("tau_moments", 12.0 * nc + 32.0 * nvo + 64.0 * nc),
        # Chebyshev step on A00 (pc_type 0): values 32 B + column 4 B per entry; rowptr 4 + 7 vectors x 16 B per row
        3: ("cheb_step_A00", 36.0 * nnzv + 116.0 * nvo),
    }
    if ctx.info(25):
        # fused AMG cycle, level 0 (SELL-64 / CSR, fp32 values: 4 B value + 4 B column per entry):
        # up-sweep x = Sb b + Sc x_c: entries of Sb and Sc, b and x per row, the coarse vector once
        kb[4] = ("amg_up0_pressure", 8.0 * ctx.info(19) + 16.0 * nvo + 8.0 * ctx.info(23))
        kb[5] = ("amg_up0_velocity_2rhs", 8.0 * ctx.info(20) + 32.0 * nvo + 16.0 * ctx.info(24))
        # down-sweep b_c = G b: entries of G, row pointer and result per coarse row, the fine vector once
        kb[8] = ("amg_down0_pressure", 8.0 * ctx.info(21) + 12.0 * ctx.info(23) + 8.0 * nvo)
        kb[9] = ("amg_down0_velocity_2rhs", 8.0 * ctx.info(22) + 20.0 * ctx.info(24) + 16.0 * nvo)
    else:
        # sweep-by-sweep cycle: level-0 Jacobi sweep (SELL-64, fp32 values): 8 B per entry; weights + 4 vectors per row
        kb[4] = ("amg_sweep_pressure", 8.0 * spnnz + 40.0 * nvo)
        kb[5] = ("amg_sweep_velocity_2rhs", 8.0 * ctx.info(8) + 56.0 * nvo)
    return kb


TIGHT = dict(snes_rtol=1e-12, snes_stol=0.0, ksp_rtol=1e-10)


def make_scenario(args, solver_name, **kw):
    """The scenario of the chosen BASELINE config on `solver_name` (a product plugin or the oracle-backed double)."""
    cfg = args.config
    if cfg == "c3":
        from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
        return DFG1Benchmark(solver_name, args.dt, 1.0, m=args.m, quiet=True, **kw)
    if cfg == "c2":
        from cfd_hemodynamic_amd.scenarios.lid_driven2D import LidDriven2DSimulation
        return LidDriven2DSimulation(solver_name, args.dt, 10.0, nx=args.nx, mu=0.01, quiet=True, **kw)
    if cfg == "c4":
        from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
        return StenosisSimulation(solver_name, args.dt, 1.0, grade="moderate", ny=args.ny, v_max=args.v_max, quiet=True, **kw)
    if cfg == "c5b":
        from cfd_hemodynamic_amd.scenarios.simple_bifurcation import MicrovasculatureSimulation
        # remove_p_mean 0: see the tolerance caveat in scenarios/simple_bifurcation.py (with the reference's mean removal the
        # steps after the first meet snes_rtol on the outlet rows alone and skip the PDE solve at this mesh scale)
        kw = dict(kw)
        kw["options"] = dict(kw.get("options", {}), remove_p_mean=int(args.remove_p_mean))
        return MicrovasculatureSimulation(solver_name, args.dt, 1.0, v_inlet=args.v_max, res=args.res3, quiet=True, **kw)
    if cfg == "p2":
        # `--solver stabilized_schur_backflow --p_grade 2` (P2/P2 triangles, do-nothing outlet + backflow term) on the
        # DFG 2D-1 channel (nu = 1e-3: tau is not at its viscous limit h^2 / 4 nu, where the P2/P2 form itself fails -- DESIGN.md section 9)
        from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
        oracle = solver_name.startswith("_oracle")  # the CPU double takes the variant as keywords (tests/oracle_solver.py)
        return DFG1Benchmark(solver_name if oracle else "stabilized_schur_backflow", args.dt, 1.0, m=args.m, v_max=0.3, p_grade=2, beta_backflow=0.2,
                             quiet=True, **(dict(kw, backflow=True) if oracle else kw))
    if cfg == "p2s":
        # the round-3 P2 workload: `--simulation stenosis --solver stabilized_schur_backflow --p_grade 2` -- does not reach its own T
        from cfd_hemodynamic_amd.scenarios.stenosis import StenosisSimulation
        oracle = solver_name.startswith("_oracle")
        return StenosisSimulation(solver_name if oracle else "stabilized_schur_backflow", args.dt, 1.0, ny=args.ny, v_max=args.v_max, p_grade=2,
                                  beta_backflow=0.2, quiet=True, **(dict(kw, backflow=True) if oracle else kw))
    if cfg == "q1h":
        # unit_cube_pipe on hexahedral cells (Q1/Q1), the reference's 213 x 4 x 4 box refined to --nx x --ny x --ny
        from cfd_hemodynamic_amd.scenarios.unit_cube_pipe import UnitCubePipeSimulation
        return UnitCubePipeSimulation(solver_name, args.dt, 1.0, p_inlet=8.85, p_outlet=0.0, nx=args.nx, ny=args.ny, nz=args.ny, quiet=True, **kw)
    if cfg == "p2t":
        # P2/P2 tetrahedra (`p_grade = 2` of stabilized_schur_backflow.py:84-87 on a 3-D mesh).  Workload: the duct
        # of unit_cube_pipe with every brick split into six tetrahedra, degree 2 in both spaces through the plugin's degree switch
        # (a measurement workload for the element type; on the millimetre-sized bifurcation P2/P2 sits in the viscous limit where
        # the Cahouet-Chabard form does not converge)
        from cfd_hemodynamic_amd.scenarios.unit_cube_pipe import UnitCubePipeSimulation
        return UnitCubePipeSimulation(solver_name, args.dt, 1.0, p_inlet=8.85, p_outlet=0.0, nx=args.nx, ny=args.ny, nz=args.ny, cell_type="tetrahedron",
                                      quiet=True, **dict(kw, _degree=2, p_grade=2))
    if cfg == "q1":
        # unit_square_pipe on quadrilateral cells (Q1/Q1), refined to --nx x --ny cells
        from cfd_hemodynamic_amd.scenarios.unit_square_pipe import UnitSquarePipeSimulation
        return UnitSquarePipeSimulation(solver_name, args.dt, 1.0, p_inlet=7.47, p_outlet=0.0, nx=args.nx, ny=args.ny, quiet=True, **kw)
    if cfg != "c5":
        raise SystemExit("no scenario for --config %s" % cfg)
    from cfd_hemodynamic_amd.scenarios.stenosis_with_tree import StenosisWithTreeSimulation
    return StenosisWithTreeSimulation(solver_name, args.dt, 1.0, grade="moderate", res=args.res, pulse_amplitude=0.5,
                                      ramp_time=args.ramp, inlet_max_velocity=args.v_max, quiet=True, **kw)


def workload_text(args, sc):
    nv = sc.solver.V.mesh.num_vertices  # nodes of the function space (= mesh vertices for P1 / Q1)
    head = {"c3": "dfg_1 (DFG 2D-1, Re=20) block-structured mesh m=%d" % args.m,
            "c2": "lid_driven2D (Re=100) unit square nx=%d" % args.nx,
            "c4": "stenosis \"moderate\" = the reference's effective geometry for every grade (L=138, R_in=1.57, R_out=1.2, x_sten=30, severity .567, slope .4), ny=%d, inlet v_max=%g mm/s, p=0 outlet" % (args.ny, args.v_max),
            "c5": "stenosis_with_tree grade moderate (L=0.03, H=0.003, severity .5, slope .5; 3-generation Murray tree, 8 outlets p=0), "
                  "res=%g, pulsatile inlet v_max (1 + 0.5 sin 2 pi t) with a (1 - cos(pi t / %g)) / 2 start-up ramp, v_max=%g" % (args.res, args.ramp, args.v_max),
            "p2": "dfg_1 channel (block mesh m=%d) with stabilized_schur_backflow --p_grade 2 --v_max 0.3: P2/P2 triangles, do-nothing outlet + backflow stabilisation" % args.m,
            "p2s": "stenosis (reference geometry) with stabilized_schur_backflow --p_grade 2: P2/P2 triangles on ny=%d cells across, inlet v_max=%g mm/s, do-nothing outlet + backflow stabilisation" % (args.ny, args.v_max),
            "q1": "unit_square_pipe (80 x 1.5 mm channel, p_inlet 7.47 / p_outlet 0, no-slip walls) on %d x %d quadrilateral cells, Q1/Q1" % (args.nx, args.ny),
            "q1h": "unit_cube_pipe (80 x 1.5 x 1.5 mm duct, p_inlet 8.85 / p_outlet 0, no-slip walls) on %d x %d x %d hexahedral cells, Q1/Q1" % (args.nx, args.ny, args.ny),
            "p2t": "unit_cube_pipe duct (80 x 1.5 x 1.5 mm, p_inlet 8.85 / p_outlet 0, no-slip walls) split into tetrahedra, %d x %d x %d bricks x 6, P2/P2" % (args.nx, args.ny, args.ny),
            "c5b": "simple_bifurcation (3-D, tetrahedra; Re=%s, inlet u_y = %g (1 - (r/r_in)^2), p = 0 at both outlets; remove_p_mean=%d), voxel-tet mesh res=%g" % (
                ("%.1f" % sc.Re) if args.config == "c5b" else "-", args.v_max, int(args.remove_p_mean), args.res3)}[args.config]
    return "%s: %d nodes, %d DOF (equal-order), dt=%g, steps from t=0, PETSc-default tolerances (snes_rtol 1e-8, ksp_rtol 1e-5)" % (
        head, nv, (sc.mesh.geometry.dim + 1) * nv, args.dt)


def run_length(args):
    """Steps of the config's own run: T = 1.0 at its dt (dfg_1.py / BASELINE configs: 100 steps at dt 0.01, 1000 at 0.001); the
    lid-driven cavity runs to T = 10."""
    return int(round((10.0 if args.config == "c2" else 1.0) / args.dt))


def step_hook(sc, k, dt):
    """Time-dependent boundary data of the step about to be solved (only config 5 has any)."""
    if hasattr(sc, "set_inlet_time"):
        sc.set_inlet_time((k + 1) * dt)


def functionals(args, sc):
    """Config-specific scalar results (global values) from a scenario whose solver offers `functional`."""
    s = sc.solver
    out = {"velocity_l2": s.functional(2), "pressure_l2": s.functional(3)}
    if args.config == "c3":
        out["drag"], out["lift"] = 500 * s.functional(0, sc.obstacle_marker), 500 * s.functional(1, sc.obstacle_marker)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process (one rank per GPU, RCCL inside libcfdh.so), pass rank 0's JSON line through, return its exit code.
    Nothing in this process has initialised the GPU (device_count() does not, on this image)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < n and os.environ.get("CFDH_SHARE_GPU") != "1":
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (CFDH_SHARE_GPU=1 rehearses the N-rank path on fewer devices)\n" % (n, ndev))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]  # "--": the launcher must not parse bench.py's options
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d%s)\n" % (n, proc.returncode, "" if line else ", no JSON line"))
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != n:
        sys.stderr.write("bench.py: the launched job reported n_gpus != %d\n" % n)
        return 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5", "c5b", "p2", "p2s", "q1", "q1h", "p2t"],
                    help="c3 (default) = the headline config; p2 / q1: the SURVEY 8f-4 element types (P2/P2 stenosis with the backflow plugin, "
                         "Q1/Q1 unit_square_pipe), no CPU leg")
    ap.add_argument("--res3", type=float, default=2.0e-4, help="c5b: voxel size of the 3-D bifurcation (2e-4: 256 k vertices, 1.03 M DOF; 1e-4: 1.95 M vertices, 7.8 M DOF)")
    ap.add_argument("--m", type=int, default=200, help="c3: DFG mesh parameter (m=200: 336,273 vertices, 1,008,819 DOF)")
    ap.add_argument("--nx", type=int, default=288, help="c2: cells per side")
    ap.add_argument("--ny", type=int, default=115, help="c4: cells across the inlet (115: 678,832 vertices, 2.04 M DOF)")
    ap.add_argument("--res", type=float, default=7.3e-6, help="c5: cell size (7.3e-6: 2.73 M vertices, 8.18 M DOF)")
    ap.add_argument("--v-max", type=float, default=None, help="inlet peak velocity (c4 default 100 mm/s, c5 default 0.05 m/s: Re = 45)")
    ap.add_argument("--ramp", type=float, default=0.03, help="c5: start-up ramp of the inlet (s); 0 = impulsive start")
    ap.add_argument("--dt", type=float, default=None, help="time step (default 0.01; c5: 0.001)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--parity-steps", type=int, default=None,
                    help="tightly converged steps compared with the oracle (default 2; c5: 0 = skipped, a tight oracle step at 8 M DOF "
                         "takes minutes -- tests/test_gpu_configs.py does that comparison on a coarse mesh of the same domain)")
    ap.add_argument("--parity-res", type=float, default=2.0e-4, help="c5: cell size of the coarse mesh of the same domain the tight parity leg runs on")
    ap.add_argument("--prof-steps", type=int, default=2, help="extra steps with HIP-event kernel timing (after the timed region)")
    ap.add_argument("--host-loop-steps", type=int, default=10, help="extra steps with the reference's literal state-copy loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--remove-p-mean", type=int, default=0, choices=[0, 1],
                    help="c5b: 1 = the reference's literal mean-pressure removal (stabilized_schur.py:319); default 0, see DESIGN.md section 9 'tolerance trap'")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host"])
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--solver", default="stabilized_schur", choices=["stabilized_schur", "stabilized_schur_bdf2"],
                    help="solver plugin (default: the headline one)")
    args = ap.parse_args()
    if args.dt is None:
        args.dt = 0.001 if args.config == "c5" else 0.01
    if args.v_max is None:
        args.v_max = {"c5": 0.05, "c5b": 1.5, "p2s": 20.0}.get(args.config, 100.0)
    if args.config == "p2s" and args.ny == 115:
        args.ny = 40      # 82 k vertices -> 330 k P2 nodes, ~1 M DOF
    if args.config == "p2" and args.m == 200:
        args.m = 80       # 54 k vertices -> 215 k P2 nodes, 0.65 M DOF (at m = 100, 1.0 M DOF, the second Newton solve of the first step
                          # needs more than one 200-vector cycle and stagnates after the restart: the P2 preconditioner's limit)
    if args.config == "q1" and (args.nx, args.ny) == (288, 115):
        args.nx, args.ny = 5870 // 2, 110 // 2   # the reference's 587 x 11 cells refined 5 x: 2935 x 55 -> 164 k nodes, 0.49 M DOF
    if args.config == "q1h" and (args.nx, args.ny) == (288, 115):
        args.nx, args.ny = 213 * 5, 4 * 2   # the reference's 213 x 4 x 4 cells: 5 x along the duct, 2 x across -> 86 k nodes, 345 k DOF
    if args.config == "p2t" and (args.nx, args.ny) == (288, 115):
        args.nx, args.ny = 213 * 2, 4       # 427 x 5 x 5 vertices -> ~80 k P2 nodes, ~320 k DOF
    if args.parity_steps is None:
        args.parity_steps = {"c5": 0, "c5b": 1, "p2t": 1, "q1h": 1}.get(args.config, 2)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves (before anything touches the GPU) and relay rank 0's line
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): refusing to report a line for a different rank count" % (world, args.gpus))

    import numpy as np
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo carries only the bootstrap (RCCL unique id) and the timing reduction;
        # halo exchange and Krylov all-reduces run on RCCL inside libcfdh.so
        # gloo announces its mesh on the C-level stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcfdh.so has no CPU fallback")
    if os.environ.get("CFDH_SHARE_GPU") == "1":
        # rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share devices; RCCL then refuses
        # the duplicate device and the library falls back to the host-staged exchange) -- never used by the driver
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    from cfd_hemodynamic_amd.parallel import PartComm

    dt = args.dt
    comm = PartComm(rank, world, args.comm) if world > 1 else None
    t0 = time.perf_counter()
    sc = make_scenario(args, args.solver, device=local_rank, comm=comm, verbose=args.verbose)
    solver = sc.solver
    ctx = solver.ctx
    t_setup = time.perf_counter() - t0
    nv = sc.mesh.num_vertices

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    its_newton, its_krylov = [], []
    solver.initStressForm()
    kstep = 0
    # wall-clock and preconditioner set-up time of every step from t = 0 on: the first step carries the build of the AMG
    # hierarchies (and the hipGraph captures), which the steady-state `value` does not show -- `end_to_end_steps_per_s` does
    step_wall, step_pc_ms = [], []
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step_hook(sc, kstep, dt)
        solver.solveStep()
        solver.assemble_wss()
        solver.advance()
        torch.cuda.synchronize()
        step_wall.append(time.perf_counter() - tw)
        step_pc_ms.append(solver.last_stats.ms_pc_setup)
        kstep += 1
    sync_all()
    ctx.profile_reset()  # zero the communication / synchronisation counters
    t0 = time.perf_counter()
    ms_asm = ms_solve = ms_pc = 0.0
    pc_rebuilds = 0
    for _ in range(args.steps):
        step_hook(sc, kstep, dt)  # time-dependent Dirichlet data (config 5), part of the step
        solver.solveStep()     # Newton + FGMRES step (stabilized_schur.py:313-334)
        solver.assemble_wss()  # per-step wall shear stress (scenario.py:262), on the device
        solver.advance()       # u_prev <- u_sol, p_prev <- p_sol (scenario.py:306-307), on the device
        kstep += 1
        st = solver.last_stats
        its_newton.append(st.newton_its)
        its_krylov.append(st.krylov_its)
        ms_asm += st.ms_assemble
        ms_solve += st.ms_solve
        ms_pc += st.ms_pc_setup
        pc_rebuilds += st.pc_refreshes
        step_pc_ms.append(st.ms_pc_setup)
    sync_all()
    elapsed = time.perf_counter() - t0
    if not step_wall:  # --warmup 0: the first timed step is the first step of the run
        step_wall.append(elapsed / args.steps + 1e-3 * step_pc_ms[0])
    counters = {k: ctx.info(i) for k, i in (("allreduce", 13), ("halo", 14), ("host_sync", 15), ("krylov", 16), ("allgather", 17), ("discarded", 73))}
    comm_size = ctx.info(18)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # The rest of the config's own run (T / dt steps from t = 0), so that the end-to-end rate is MEASURED, not extrapolated from the
    # timed region: with the projected initial guess the steps get cheaper as the flow settles.  Bounded (20 s); every step so far
    # was clocked, the first one with its preconditioner build and graph captures included.
    n_run_total = run_length(args)
    steps_before = kstep
    t_rest0 = time.perf_counter()
    rest_failure = None
    while world == 1 and kstep < n_run_total and time.perf_counter() - t_rest0 < 20.0:
        try:
            step_hook(sc, kstep, dt)
            solver.solveStep()
            solver.assemble_wss()
            solver.advance()
        except RuntimeError as e:  # a run that does not reach its own T (the P2 stenosis: DESIGN.md section 9) is reported, not fatal
            rest_failure = "step %d: %s" % (kstep + 1, str(e)[:160])
            break
        kstep += 1
    torch.cuda.synchronize()
    t_rest = time.perf_counter() - t_rest0
    if rest_failure:  # the state is that of a failed step: no further legs on this scenario
        args.host_loop_steps = 0
        args.prof_steps = 0
    e2e_steps = kstep
    e2e_wall = sum(step_wall[:args.warmup]) + elapsed + t_rest
    rest_rate = (kstep - steps_before) / t_rest if kstep > steps_before else None

    # the reference's literal loop: u_prev.x.array[:] = u_sol.x.array[:] (scenario.py:306-307), reported beside `value`
    literal_rate = None
    if world == 1 and args.host_loop_steps > 0:
        sync_all()
        t0h = time.perf_counter()
        for _ in range(args.host_loop_steps):
            step_hook(sc, kstep, dt)
            solver.solveStep()
            solver.assemble_wss()
            solver.u_prev.x.array[:] = solver.u_sol.x.array[:]
            solver.p_prev.x.array[:] = solver.p_sol.x.array[:]
            kstep += 1
        sync_all()
        literal_rate = args.host_loop_steps / (time.perf_counter() - t0h)

    results = functionals(args, sc)

    # Kernel durations: HIP events on the library's stream around every launch of the hot kernels, over `prof_steps`
    # further steps of the same run (the preconditioner's hipGraph replay is switched off while events are recorded;
    # the kernels and their data are the same)
    kern, roof = [], None
    if args.prof_steps > 0:
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(args.prof_steps):
            step_hook(sc, kstep, dt)
            solver.solveStep()
            solver.advance()
            kstep += 1
        ctx.profile_enable(False)
        sync_all()
        # An empty event pair on the stream already reads ~4.5 us (kind 7, recorded when profiling is switched on); a
        # pair around a kernel overlaps part of that with the launch, so the excess over rocprofv3's kernel-trace
        # duration is ~3 us per launch.  `avg_us` is the RAW event time (the achieved rates are lower bounds).
        oms, on = ctx.profile_get(7)
        ovh_us = 1e3 * oms / on if on else 0.0
        for kind, (name, nbytes) in kernel_bytes(ctx).items():
            ms, n = ctx.profile_get(kind)
            if n:
                raw = 1e3 * ms / n
                kern.append({"kernel": name, "launches": n, "avg_us": raw, "total_ms": ms, "algorithmic_MB": nbytes / 1e6,
                             "GBps": nbytes / (raw * 1e-6) / 1e9, "avg_us_minus_empty_event_pair": max(raw - ovh_us, 0.0)})
        kern.sort(key=lambda k: -k["total_ms"])
        if kern:
            d = kern[0]
            # HBM traffic of the PMC counters: collected by tools/profile_pmc.sh in separate rocprofv3 --pmc passes (it cannot be
            # measured inside this run) and accepted only if it was taken with EXACTLY the kernel sources of this run
            traffic, traffic_note = None, "no PMC summary for this workload"
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and args.config == "c3" and args.m == 200:
                try:
                    rec = json.load(open(pmc))
                    if rec.get("kernel_source_sha16") == kernel_source_sha16():
                        traffic = rec.get("per_launch_bytes", {}).get(d["kernel"])
                        traffic_note = "profiles/pmc_traffic.json, kernel sources %s" % rec.get("kernel_source_sha16")
                    else:
                        traffic_note = "profiles/pmc_traffic.json is stale (taken with kernel sources %s, this run has %s): not reported" % (
                            rec.get("kernel_source_sha16"), kernel_source_sha16())
                except Exception as e:
                    traffic_note = "profiles/pmc_traffic.json unreadable: %s" % e
            roof = {"bound": "hbm", "kernel": d["kernel"], "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note, "avg_us": d["avg_us"],
                    "algorithmic_bytes": d["algorithmic_MB"] * 1e6, "empty_event_pair_us": ovh_us}

    label = {"c3": "dfg_1 ~1M DOF", "c2": "lid_driven2D ~250k DOF", "c4": "stenosis moderate ~2M DOF",
             "c5": "stenosis_with_tree ~8M DOF pulsatile", "c5b": "simple_bifurcation 3-D tets", "p2": "dfg_1 P2/P2 ~1M DOF (backflow plugin)", "p2s": "stenosis P2/P2 ~1M DOF (backflow plugin)",
             "q1": "unit_square_pipe Q1/Q1 quadrilaterals", "q1h": "unit_cube_pipe Q1/Q1 hexahedra",
             "p2t": "unit_cube_pipe duct on P2/P2 tetrahedra"}[args.config]
    kits = max(sum(its_krylov), 1)
    out = {
        "metric": "time-steps/sec, %s (%s)" % (label, args.solver),
        "value": args.steps / elapsed,
        "unit": "time-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64 (fp32 only in the AMG matrix values of the flexible preconditioner and in a read-only copy of the Krylov basis that Gram-Schmidt reads in solves of >= 20 iterations)",
        "data": "synthetic",
        "config": {"workload": workload_text(args, sc),
                   "parallelism": "element partition x%d (RCB), halo + dot all-reduce on %s; communicator size %d" % (
                       world, (comm.backend if comm is not None else args.comm) + (
                           " [fallback: %s]" % comm.fallback_reason if getattr(comm, "fallback_reason", None) else ""), comm_size)},
        "ms_assemble_per_step": ms_asm / args.steps,
        "ms_solve_per_step": ms_solve / args.steps,
        "ms_pc_setup_per_step": ms_pc / args.steps,
        "pc_hierarchy_rebuilds_in_timed_steps": int(pc_rebuilds),  # lagged AMG hierarchies: host rebuilds, inside the timed region
        "newton_its_per_step": float(np.mean(its_newton)),
        "krylov_its_per_step": float(np.mean(its_krylov)),
        "per_krylov_iteration": {"allreduce": counters["allreduce"] / kits, "halo_exchange": counters["halo"] / kits,
                                 "allgather": counters["allgather"] / kits, "host_sync": counters["host_sync"] / kits,
                                 # FGMRES iterations launched ahead of the host's convergence test and thrown away (not in krylov_its)
                                 "launched_ahead_and_discarded": counters["discarded"] / kits},
        # initial guess of the linear solves (cfdh_options.ksp_guess, PETSc's KSPGuess): each solve starts from the projection of its
        # right-hand side onto the solutions of the same Newton solve of the last steps; every solve is still run to rtol |b|
        "linear_solver_initial_guess": {"ksp_guess": int(sc.solver.options.ksp_guess), "solves_with_a_projected_guess": int(sc.solver.ctx.info(70)),
                                        "mean_initial_residual_over_rhs": 1e-6 * sc.solver.ctx.info(71)},
        # FGMRES solves that the attainable-accuracy rule ended ABOVE their tolerance (reason CFDH_KSP_CONVERGED_ATTAINABLE, cfdh_info 72;
        # DESIGN.md section 6): 0 means every solve of this context so far met rtol |b| on the true residual
        "solves_stopped_at_attainable_accuracy": int(ctx.info(72)),
        "pressure_level1_rows": int(ctx.info(23)),  # rows of level 1 of the pressure hierarchy: the coarse right-hand side a partitioned run all-reduces
        "setup_s": t_setup,
        # the reference's literal loop `u_prev.x.array[:] = u_sol.x.array[:]` (scenario.py:306-307): the lazy array proxy maps
        # that idiom to a device copy, so no field crosses PCIe in it
        "literal_reference_loop_steps_per_s": literal_rate,
        "results": results,
        "roofline": roof,
        "kernels": kern,
    }
    # The config's own run, end to end: T / dt steps from t = 0 INCLUDING the first step's preconditioner build (measured:
    # wall-clock of step 1 of this run) -- next to the steady-state `value`.  `hierarchy_build_s` = preconditioner set-up
    # time of all steps before the timed region (the AMG hierarchies are built in step 1 and lagged after).
    n_run = run_length(args)
    t_first = step_wall[0]
    if world == 1 and (e2e_steps > args.warmup + args.steps or rest_failure):
        out["end_to_end_measured"] = {"steps": int(e2e_steps), "of_run_length": int(n_run), "wall_s": e2e_wall,
                                      "steps_per_s": e2e_steps / e2e_wall, "steps_per_s_after_the_timed_region": rest_rate,
                                      "note": "every step from t = 0 clocked (first step with its preconditioner build included); context creation excluded"}
        if rest_failure:
            out["end_to_end_measured"]["stopped"] = rest_failure
    out["hierarchy_build_s"] = 1e-3 * sum(step_pc_ms[:max(args.warmup, 1)])
    out["first_step_s"] = t_first
    out["run_length_steps"] = n_run
    # extrapolation of the timed region to the run length -- only for a run that is known to reach its own T: a run that stopped
    # early in the clocked rest-of-run loop above gets null here (its `end_to_end_measured.stopped` says where)
    if rest_failure:
        out["end_to_end_steps_per_s"] = out["end_to_end_steps_per_s_incl_setup"] = None
    else:
        out["end_to_end_steps_per_s"] = n_run / (t_first + (n_run - 1) * elapsed / args.steps)
        out["end_to_end_steps_per_s_incl_setup"] = n_run / (t_setup + t_first + (n_run - 1) * elapsed / args.steps)
    if args.config == "c3":
        out["drag_coefficient"], out["lift_coefficient"], out["velocity_l2"] = results["drag"], results["lift"], results["velocity_l2"]

    if args.config == "c5b":
        qi, q1, q2 = sc.flow_rates()
        out["results"].update({"inflow": qi, "outflow_1": q1, "outflow_2": q2})
    if world == 1 and rank == 0 and int(sc.solver.options.ksp_guess) > 0 and args.config in ("c2", "c3", "c4", "c5b", "q1", "q1h"):
        # The same timed region with the linear solver configured like the reference's KSP: zero initial guess in every solve (and
        # the fp64 basis throughout) -- so that the line carries both numbers and the share of `value` that is due to the projected
        # guess can be read off.  A fresh scenario from t = 0, the same warm-up and step counts, the same clock.
        fp32_before = os.environ.get("CFDH_KRYLOV_FP32")
        os.environ["CFDH_KRYLOV_FP32"] = "0"
        sc0 = make_scenario(args, args.solver, device=local_rank, options=dict(ksp_guess=0))
        k0 = 0
        for _ in range(args.warmup):
            step_hook(sc0, k0, dt); sc0.solver.solveStep(); sc0.solver.assemble_wss(); sc0.solver.advance(); k0 += 1
        torch.cuda.synchronize()
        t00 = time.perf_counter()
        its0 = 0
        for _ in range(args.steps):
            step_hook(sc0, k0, dt); sc0.solver.solveStep(); sc0.solver.assemble_wss(); sc0.solver.advance(); k0 += 1
            its0 += sc0.solver.last_stats.krylov_its
        torch.cuda.synchronize()
        e0 = time.perf_counter() - t00
        if fp32_before is None:
            os.environ.pop("CFDH_KRYLOV_FP32", None)
        else:
            os.environ["CFDH_KRYLOV_FP32"] = fp32_before
        out["zero_initial_guess_check"] = {"ksp_guess": 0, "steps_per_s": args.steps / e0, "ms_per_step": 1e3 * e0 / args.steps,
                                           "krylov_its_per_step": its0 / args.steps,
                                           "note": "same timed region, linear solves started from zero as the reference's KSP does"}
        del sc0
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # CPU baseline and parity: the same Scenario class on the oracle-backed test double of the solver plugin
        # (tests/oracle_solver.py over oracle/cfdh_oracle.c) -- checker and reported baseline only, never the product.
        import oracle_solver
        mod = types.ModuleType("cfd_hemodynamic_amd.solvers._oracle_double")
        mod.Solver = oracle_solver.Solver
        sys.modules["cfd_hemodynamic_amd.solvers._oracle_double"] = mod
        cores = host_cores()  # all host cores this process may use (stated in the line)
        os.environ["CFDH_ORACLE_THREADS"] = str(cores)
        bdf2 = args.solver == "stabilized_schur_bdf2"
        # same Newton / FGMRES / Cahouet-Chabard + AMG algorithm and tolerances; FULL Schur factorisation, which is
        # the faster variant on the CPU (4.0 vs 2.7 steps/s with the upper-triangular factor the GPU path prefers)
        # ... and the projected initial guess of the linear solves (orc_opts.ksp_guess = cfdh_options.ksp_guess) when that is the
        # faster way on the CPU too: the sample is run with the GPU path's setting and with the zero guess, the FASTER one is reported
        # (at 1 M DOF the four extra products cost the CPU port more than the iterations they save; elsewhere they pay)
        g_gpu = int(sc.solver.options.ksp_guess)

        def cpu_leg(guess):
            o_sc = make_scenario(args, "_oracle_double", pc_kind=2, options=dict(schur_upper=0, ksp_guess=guess), bdf2=bdf2)
            o_sc.solver.O.set_threads(cores)
            t_start = time.perf_counter()
            n_done = 0
            for _ in range(args.warmup + args.cpu_steps):
                step_hook(o_sc, n_done, dt)
                o_sc.solver.solveStep()
                o_sc.solver.advance()
                n_done += 1
                if n_done == args.warmup:
                    t_start = time.perf_counter()
                if time.perf_counter() - t_start > 60.0 and n_done > args.warmup:
                    break
            return o_sc, n_done, n_done - args.warmup, time.perf_counter() - t_start

        osc, nst, ncpu, tcpu = cpu_leg(g_gpu)
        legs = {g_gpu: ncpu / tcpu}
        if g_gpu > 0:
            o0, nst0, ncpu0, tcpu0 = cpu_leg(0)
            legs[0] = ncpu0 / tcpu0
            del o0
            if legs[0] > legs[g_gpu]:
                ncpu, tcpu = ncpu0, tcpu0
        best = max(legs, key=legs.get)
        out["cpu_baseline"] = {
            "value": ncpu / tcpu, "unit": "time-steps/s", "cores": cores, "kind": "port", "host_cores_available": host_cores(),
            "sample": "steps %d..%d of the same mesh/dt from t=0 with the C oracle (oracle/cfdh_oracle.c, pc_kind=2: "
                      "same Newton + FGMRES + Cahouet-Chabard/AMG preconditioner and tolerances, FULL Schur factorisation = "
                      "the faster variant on the CPU, OpenMP; initial guess of the linear solves: the faster of %s, here ksp_guess = %d)" % (
                          args.warmup + 1, nst, " / ".join("ksp_guess = %d: %.3f steps/s" % (k, v) for k, v in sorted(legs.items())), best),
            "ms_per_step": 1e3 * tcpu / max(ncpu, 1),
        }
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]

        # Parity at the bench size (drag/lift: dfg_1.py:183-211, L2 norms: scenario.py:315-324).
        rel = lambda a, b: abs(a - b) / abs(b) if b != 0 else abs(a - b)

        def compare(sc_g, sc_o):
            fg, fo = functionals(args, sc_g), functionals(args, sc_o)
            xg = np.concatenate([np.asarray(sc_g.solver.u_sol.x.array), np.asarray(sc_g.solver.p_sol.x.array)])
            xo = sc_o.solver.x_n
            d = {"%s_rel" % k: rel(fg[k], fo[k]) for k in fo}
            d["l2_rel"] = d["velocity_l2_rel"]
            d["solution_rel"] = float(np.linalg.norm(xg - xo) / np.linalg.norm(xo))
            d["gpu"], d["oracle"] = fg, fo
            return d

        # (1) the states both sides reached at PETSc-default tolerances after `nst` steps: differences = solver noise
        sc2 = make_scenario(args, args.solver, device=local_rank)
        for k in range(nst):
            step_hook(sc2, k, dt)
            sc2.solver.solveStep()
            sc2.solver.advance()
        out["parity_default_tolerances"] = dict(compare(sc2, osc), step=nst)
        del sc2, osc
        # (2) steps from rest with BOTH sides converged tightly: this is the `parity` entry, north_star's
        #     "drag/lift within 1e-6 relative" is read against it
        if args.parity_steps > 0:
            sc3 = make_scenario(args, args.solver, device=local_rank, options=dict(TIGHT))
            osc3 = make_scenario(args, "_oracle_double", pc_kind=2, options=dict(TIGHT), bdf2=bdf2)
            osc3.solver.O.set_threads(cores)
            for k in range(args.parity_steps):
                for s_ in (sc3, osc3):
                    step_hook(s_, k, dt)
                    s_.solver.solveStep()
                    s_.solver.advance()
            out["parity"] = dict(compare(sc3, osc3), step=args.parity_steps,
                                 tolerances="steps from rest, both sides snes_rtol 1e-12 / ksp_rtol 1e-10 (oracle pc_kind=2)")
            del sc3, osc3
        elif args.config == "c5":
            # a tight oracle step at 8 M DOF takes minutes: the tight comparison of this config runs on a coarse mesh of the
            # SAME domain, boundary data and dt (same scenario class, res = --parity-res)
            import copy
            ca = copy.copy(args)
            ca.res = args.parity_res
            sc3 = make_scenario(ca, args.solver, device=local_rank, options=dict(TIGHT))
            osc3 = make_scenario(ca, "_oracle_double", pc_kind=2, options=dict(TIGHT), bdf2=bdf2)
            osc3.solver.O.set_threads(cores)
            for k in range(3):
                for s_ in (sc3, osc3):
                    step_hook(s_, k, dt)
                    s_.solver.solveStep()
                    s_.solver.advance()
            out["parity"] = dict(compare(sc3, osc3), step=3, mesh="coarse mesh of the same domain: res=%g, %d vertices" % (ca.res, sc3.mesh.num_vertices),
                                 tolerances="steps from rest, both sides snes_rtol 1e-12 / ksp_rtol 1e-10 (oracle pc_kind=2)")
            del sc3, osc3

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
