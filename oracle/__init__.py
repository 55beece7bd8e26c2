"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/gpe_oracle.h)."""
