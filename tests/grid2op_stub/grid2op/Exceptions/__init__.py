class Grid2OpException(RuntimeError):
    pass


class BackendError(Grid2OpException):
    pass
