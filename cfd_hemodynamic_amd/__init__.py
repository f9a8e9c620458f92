"""MI355X-native implementation of the reference's `stabilized_schur` time step
behind its Scenario / SolverBase / BoundaryCondition plugin surface."""
__version__ = "0.1.0"
