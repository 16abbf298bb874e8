class OPFNotConverged(Exception):
    pass
