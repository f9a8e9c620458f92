"""`Scenario` ABC + the time loop that calls the hot path, with the reference's
semantics (/root/reference/src/scenario.py:20-360): solver loaded by name,
kwargs filtered by the constructor signature, `setup()`, `solve()` with the
float-accumulated `while t < T`, the every-10th-step early stop evaluated before
the prev-copy, final L2 norms written to norms.txt.

Differences, all outside the hot path: no ADIOS2/VTX output (results are kept in
memory and, optionally, written as .npz snapshots), no tqdm bar.  When the
solver offers `advance()`/`functional()` the loop keeps the fields in HBM
(`device_resident=True`, default) instead of copying `*.x.array` through the
host every step; `device_resident=False` runs the reference's literal loop.
"""
from __future__ import annotations

import inspect
import os
import time
from abc import ABC, abstractmethod
from importlib import import_module
from typing import Callable

import numpy as np

from .boundaryCondition import BoundaryCondition
from .fem import Function
from .solverBase import SolverBase


class Scenario(ABC):
    @property
    @abstractmethod
    def mesh(self):
        pass

    @property
    @abstractmethod
    def bcu(self) -> list[BoundaryCondition]:
        pass

    @property
    @abstractmethod
    def bcp(self) -> list[BoundaryCondition]:
        pass

    @abstractmethod
    def initial_velocity(self, x: np.ndarray) -> np.ndarray:
        pass

    def exact_velocity(self, t):
        pass

    def __init__(self, solver_name: str, scenario_name: str, rho: float, mu: float, dt: float, T: float, f: list,
                 early_stop_tolerance: float = 1e-3, **solver_kwargs):
        """Loads `solvers/<solver_name>.py`, keeps the keyword arguments its `Solver.__init__` can take
        (all of them when it declares **kwargs) and builds the solver on `self.mesh` -- the plugin
        protocol of /root/reference/src/scenario.py:44-133 (ImportError: no such module, ValueError: module
        without a `Solver`, RuntimeError: the constructor failed)."""
        self.solver_name, self.scenario_name = solver_name, scenario_name
        self.early_stop_tolerance = early_stop_tolerance
        self.dt, self.T = dt, T
        self.solverClass: type[SolverBase] = self._find_solver_class(solver_name)
        self.solver = self._build_solver(self.solverClass, (self.mesh, dt, rho, mu, f), solver_kwargs)
        self.has_exact_solution = type(self).exact_velocity is not Scenario.exact_velocity
        self.step_stats = []

    @staticmethod
    def _solver_names() -> list[str]:
        """Plugin modules present next to this file."""
        folder = os.path.join(os.path.dirname(os.path.abspath(__file__)), "solvers")
        if not os.path.isdir(folder):
            return []
        return sorted(n[:-3] for n in os.listdir(folder) if n.endswith(".py") and n[0] != "_")

    @classmethod
    def _find_solver_class(cls, name: str):
        try:
            module = import_module(f"{__package__}.solvers.{name}")
        except ImportError as exc:
            known = ", ".join(cls._solver_names()) or "none"
            raise ImportError(f"no solver plugin '{name}' could be imported from {__package__}/solvers "
                              f"({exc}); plugins present: {known}") from exc
        solver_class = getattr(module, "Solver", None)
        if solver_class is None:
            raise ValueError(f"{__package__}/solvers/{name}.py has no class named Solver")
        return solver_class

    def _build_solver(self, solver_class, positional, keywords: dict):
        params = inspect.signature(solver_class.__init__).parameters
        if not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
            keywords = {k: v for k, v in keywords.items() if k in params}  # silently dropped, as in the reference
        try:
            return solver_class(*positional, initial_velocity=self.initial_velocity, **keywords)
        except TypeError as exc:
            raise RuntimeError(f"solver '{self.solver_name}' rejected its constructor arguments: {exc}") from exc
        except Exception as exc:
            raise RuntimeError(f"solver '{self.solver_name}' failed during construction "
                               f"({type(exc).__name__}: {exc})") from exc

    @property
    def facet_tags(self):
        return getattr(self, "_ft", None)

    @property
    def tags(self) -> dict:
        return {
            "inlet": getattr(self, "inlet_marker", None),
            "outlet": getattr(self, "outlet_marker", None),
            "wall": getattr(self, "wall_marker", None),
            "obstacle": getattr(self, "obstacle_marker", None),
        }

    def setup(self):
        self.solver.setup(self.bcu, self.bcp, facet_tags=self.facet_tags, tags=self.tags)
        if self.mesh.comm.rank == 0 and not getattr(self, "quiet", False):
            nV = self.solver.V.dofmap.index_map.size_global * self.solver.V.dofmap.index_map_bs
            nQ = self.solver.Q.dofmap.index_map.size_global * self.solver.Q.dofmap.index_map_bs
            print(f"DOFs: {nV + nQ} (Velocity: {nV}, Pressure: {nQ})")
            print(f"Suggested cores: {(nV + nQ) / 20000:.1f}")

    def solve(self, output_folder: str = None, afterStepCallback: Callable[[float], None] = None,
              device_resident: bool = True, max_steps: int = None, write_every: int = 1) -> str:
        mesh, T, solver = self.mesh, self.T, self.solver
        quiet = getattr(self, "quiet", False)
        if output_folder and mesh.comm.rank == 0:
            os.makedirs(output_folder, exist_ok=True)
        mesh.comm.barrier()
        solver.initStressForm()
        # v / p / u_residual / p_residual / wss time series (scenario.py:208-228): VTU + PVD instead of ADIOS2 VTX.
        # write_every = 1 is the reference's behaviour (every step); 0 disables the series.
        writers = []
        if output_folder and write_every:
            from .io import VTUWriter
            writers = [VTUWriter(mesh.comm, f"{output_folder}/v.bp", solver.u_sol, "v"),
                       VTUWriter(mesh.comm, f"{output_folder}/p.bp", solver.p_sol, "p"),
                       VTUWriter(mesh.comm, f"{output_folder}/u_residual.bp", solver.u_residual, "u_residual"),
                       VTUWriter(mesh.comm, f"{output_folder}/p_residual.bp", solver.p_residual, "p_residual"),
                       VTUWriter(mesh.comm, f"{output_folder}/wss.bp", solver.shear_stress, "shear_stress")]
        t = 0.0
        solver.u_sol.interpolate(self.initial_velocity)
        if writers:
            solver.assemble_wss()
            for w in writers:
                w.write(t)
        error_log = None
        self.errors = []  # (t, relative L2 velocity error) when the scenario has an exact solution
        if self.has_exact_solution:
            error_log = open(f"{output_folder}/err.txt", "w") if (output_folder and mesh.comm.rank == 0) else None
            u_e = Function(solver.V)
            u_e.interpolate(lambda x: self.exact_velocity(t)(x))
            error = self.compute_error(solver.u_sol, u_e, mesh)
            self.errors.append((t, error))
            if error_log:
                error_log.write("t = %.3f: error = %.3g" % (t, error) + "\n")
        fast = device_resident and hasattr(solver, "advance") and hasattr(solver, "functional")
        i = 0
        self.step_stats = []
        self.stopped_early = False
        while t < T:
            t0 = time.perf_counter()
            solver.solveStep()
            st = getattr(solver, "last_stats", None)
            self.step_stats.append((time.perf_counter() - t0, st))
            i += 1
            t += self.dt
            if self.has_exact_solution:
                u_e.interpolate(self.exact_velocity(t))
                error = self.compute_error(u_e, solver.u_sol, mesh)
                self.errors.append((t, error))
                if error_log:
                    error_log.write("t = %.3f: error = %.3g" % (t, error) + "\n")
            solver.assemble_wss()
            if writers and i % write_every == 0:
                for w in writers:
                    w.write(t)
            if afterStepCallback:
                afterStepCallback(t)
            if (i + 1) % 10 == 0:
                if fast:
                    u_sol_norm = solver.functional(4)
                    u_prev_norm = solver.functional(5)
                    u_diff_norm = solver.functional(6)
                else:
                    u_sol_arr = solver.u_sol.x.array
                    u_prev_arr = solver.u_prev.x.array
                    u_sol_norm = mesh.comm.allreduce(np.linalg.norm(u_sol_arr, ord=np.inf))
                    u_prev_norm = mesh.comm.allreduce(np.linalg.norm(u_prev_arr, ord=np.inf))
                    u_diff_norm = mesh.comm.allreduce(np.linalg.norm(u_sol_arr - u_prev_arr, ord=np.inf))
                rel_diff = (u_diff_norm / max(u_sol_norm, 1e-12)) / self.dt
                if mesh.comm.rank == 0 and not quiet:
                    print(f"Step {i+1}: t={t:.3f} ||u_sol||={u_sol_norm:.6e}, ||u_prev||={u_prev_norm:.6e}, "
                          f"||diff||={u_diff_norm:.6e} rel_diff/dt={rel_diff:.6e}")
                if rel_diff < self.early_stop_tolerance:
                    if mesh.comm.rank == 0 and not quiet:
                        print(f"Early stopping at t={t:.3f}, because (||u_sol - u_prev||_inf / ||u_sol||_inf) / dt = "
                              f"{rel_diff:.20e} < {self.early_stop_tolerance}")
                    self.stopped_early = True
                    break
            if fast:
                solver.advance()
            else:
                solver.u_prev.x.array[:] = solver.u_sol.x.array[:]
                solver.p_prev.x.array[:] = solver.p_sol.x.array[:]
            if max_steps is not None and i >= max_steps:
                break
        self.num_steps = i
        self.t_end = t
        if fast:
            norm_v, norm_p = solver.functional(2), solver.functional(3)
        else:
            norm_v, norm_p = self._l2_norms_host()
        self.norm_v, self.norm_p = norm_v, norm_p
        solver.assemble_wss()
        if output_folder:
            # reading `x.array` gathers the owned slices in a partitioned run: every rank takes part, rank 0 writes
            fields = {k: np.asarray(f.x.array) for k, f in (('velocity', solver.u_sol), ('pressure', solver.p_sol), ('wss', solver.shear_stress))}
            if mesh.comm.rank == 0:
                with open(os.path.join(output_folder, "norms.txt"), "w") as f:
                    f.write(f"L2 norm of velocity: {norm_v}\n")
                    f.write(f"L2 norm of pressure: {norm_p}\n")
                np.savez(os.path.join(output_folder, "final.npz"), x=mesh.x, cells=mesh.cells, **fields)
            mesh.comm.barrier()
        for w in writers:
            w.close()
        if error_log:
            error_log.close()
        return output_folder

    @staticmethod
    def _mass_weights(mesh):
        """int_K l_a l_b = |K| (1 + d_ab) / ((d+1)(d+2)) per cell."""
        n1 = mesh.cells.shape[1]
        vol = mesh.cell_areas() if n1 == 3 else mesh.cell_volumes()
        return vol[:, None, None] * (1.0 + np.eye(n1))[None] / (n1 * (n1 + 1.0))

    def _l2_norms_host(self):
        m = self.mesh
        mab = self._mass_weights(m)
        ue = np.asarray(self.solver.u_sol.x.array).reshape(-1, m.geometry.dim)[m.cells]
        pe = np.asarray(self.solver.p_sol.x.array)[m.cells]
        return (float(np.sqrt(np.einsum("cab,cai,cbi->", mab, ue, ue))),
                float(np.sqrt(np.einsum("cab,ca,cb->", mab, pe, pe))))

    @staticmethod
    def compute_error(u: Function, u_aprox: Function, mesh) -> float:
        """Relative L2 error (/root/reference/src/scenario.py:350-360)."""
        mab = Scenario._mass_weights(mesh)
        bs = u.function_space.bs
        a = u.x.array.reshape(-1, bs)[mesh.cells]
        b = u_aprox.x.array.reshape(-1, bs)[mesh.cells]
        d = b - a
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.sqrt(np.einsum("cab,cai,cbi->", mab, d, d)) / np.sqrt(np.einsum("cab,cai,cbi->", mab, a, a)))
