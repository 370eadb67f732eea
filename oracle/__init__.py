"""CPU oracle package (test infrastructure only; see fov_oracle.py header)."""
