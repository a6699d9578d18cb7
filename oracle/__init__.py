"""CPU oracle (test infrastructure only -- never imported by the product package)."""
from .oracle import *  # noqa: F401,F403
