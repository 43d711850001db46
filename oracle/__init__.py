"""TEST INFRASTRUCTURE ONLY: fp64 CPU restatement used as the parity checker (see pih_oracle.h)."""
