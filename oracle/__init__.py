"""CPU oracle of the env spec - test infrastructure only (see racecar_oracle.py header)."""
