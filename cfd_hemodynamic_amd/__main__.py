"""`python -m cfd_hemodynamic_amd simulate --simulation dfg_1 --solver stabilized_schur --T 1.0 --dt 0.01 --name X`
-- the `simulate` sub-command of the reference's CLI (/root/reference/main.py:87-139,253-254);
unknown `--key value` pairs are literal-evaluated and passed to the scenario (main.py:12-31)."""
from __future__ import annotations

import argparse
import ast
import inspect
import os
import sys
from importlib import import_module

from .scenario import Scenario


def _parse_extra(extra):
    out = {}
    it = iter(extra)
    for k in it:
        if not k.startswith("--"):
            raise ValueError(f"unexpected argument {k}")
        v = next(it, None)
        if v is None:
            raise ValueError(f"missing value for {k}")
        try:
            out[k[2:]] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k[2:]] = v
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cfd_hemodynamic_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("simulate")
    s.add_argument("--simulation", required=True)
    s.add_argument("--solver", default="stabilized_schur")
    s.add_argument("--T", type=float, required=True)
    s.add_argument("--dt", type=float, required=True)
    s.add_argument("--name", default="run")
    s.add_argument("--output_dir", default="results")
    args, extra = ap.parse_known_args(argv)
    kw = _parse_extra(extra)
    try:
        mod = import_module(f"{__package__}.scenarios.{args.simulation}")
    except ImportError as e:
        raise ImportError(f"unknown simulation '{args.simulation}': {e}") from e
    cls = next(c for _, c in inspect.getmembers(mod, inspect.isclass) if issubclass(c, Scenario) and c is not Scenario)
    sim = cls(args.solver, args.dt, args.T, **kw)
    out = os.path.join(args.output_dir, args.simulation, args.name)
    sim.setup()  # the reference calls setup() a second time in Simulation.run (simulation.py:269)
    sim.solve(out)
    print("results in", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
