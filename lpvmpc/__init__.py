"""Importable alias of the package directory ``autonomous-racing-lpv-mpp-mpc_amd/`` (whose name is not a
valid Python identifier).  All code lives there; this file only points ``__path__`` at it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "autonomous-racing-lpv-mpp-mpc_amd")]

from .api import (BatchedSolver, LPV_MPC_Planner, PathFollowingLPV_MPC, PlannerHandoff,  # noqa: E402,F401
                  body_frame_errors, handoff_operators)
from ._ffi import LpvMpcError, STATUS_TEXT  # noqa: E402,F401
from .track import Map  # noqa: E402,F401

__all__ = ["BatchedSolver", "PathFollowingLPV_MPC", "LPV_MPC_Planner", "PlannerHandoff", "body_frame_errors", "handoff_operators",
           "Map", "LpvMpcError", "STATUS_TEXT"]
