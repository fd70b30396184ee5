"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/avatar_oracle.cpp header)."""
