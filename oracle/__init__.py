"""CPU oracle of the screening hot path. TEST INFRASTRUCTURE ONLY (see pmx_oracle.c header)."""
