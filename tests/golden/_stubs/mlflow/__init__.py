"""No-op experiment tracking."""
from . import pyfunc  # noqa: F401


def log_params(*_, **__):
    return None


def log_metrics(*_, **__):
    return None


def set_experiment(*_, **__):
    return None


def active_run():
    return None


def start_run(*_, **__):
    return None


def end_run(*_, **__):
    return None
