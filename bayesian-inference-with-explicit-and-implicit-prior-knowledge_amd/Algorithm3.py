"""Mirror of reference src/Algorithm3.py: the class lives in Algorithm1.py next to its base class (as the reference's
Algorithm3 derives from Algorithm1); this module keeps the reference's import path."""
from .Algorithm1 import Algorithm3  # noqa: F401
