"""CPU oracle (test infrastructure only) -- see oracle/tm_oracle.py header."""
from . import tm_oracle  # noqa: F401
