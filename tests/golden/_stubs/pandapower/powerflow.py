from oracle.pf_oracle import LoadflowNotConverged  # noqa: F401
