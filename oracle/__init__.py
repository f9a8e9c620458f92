"""CPU oracle (test infrastructure only).  See cfdh_oracle.c / np_twin.py headers."""
