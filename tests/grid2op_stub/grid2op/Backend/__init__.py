from grid2op.Backend.backend import Backend  # noqa: F401
