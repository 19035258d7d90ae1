from .logger import setup_logging  # noqa: F401
