"""Import alias: `ofdm_course_amd` -> the package directory `ofdm-course_amd/` (a hyphen is not
importable).  All code lives in `ofdm-course_amd/`; this file only extends the search path."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ofdm-course_amd"))

from .api import *  # noqa: F401,F403,E402
from . import api as api  # noqa: E402
