"""CPU oracle for the ICP hot path -- TEST INFRASTRUCTURE ONLY (see icp_oracle.c)."""
