from .prediction import predict  # noqa: F401
