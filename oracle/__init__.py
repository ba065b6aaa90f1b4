# TEST INFRASTRUCTURE ONLY (see oracle/lm_oracle.hpp header).
