"""The non-executing pickle reader lives in the package (trajopt_amd/static_pickle.py); the fixture scripts
in this directory import it under its old name."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trajopt_amd.static_pickle import Global, Inert, decode, read, walk  # noqa: E402,F401
