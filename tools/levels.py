"""Prints the AMG hierarchies of the bench workload (level sizes / nnz), one step with verbose=1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_hemodynamic_amd.scenarios.dfg_1 import DFG1Benchmark
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = DFG1Benchmark("stabilized_schur", 0.01, 1.0, m=m, quiet=True, verbose=1)
sc.solver.solveStep()
