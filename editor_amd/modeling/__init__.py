from .make_model import make_model  # noqa: F401
