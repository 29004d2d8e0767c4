"""MI355X-native batched LPV-MPC / LPV-MPP solve path (drop-in for the reference's
PathFollowingLPV_MPC / LPV_MPC_Planner classes).  Import as ``lpvmpc`` (alias package at the repo root)
or via ``importlib.import_module("autonomous-racing-lpv-mpp-mpc_amd")``."""
from .api import (BatchedSolver, LPV_MPC_Planner, PathFollowingLPV_MPC, PlannerHandoff,  # noqa: F401
                  body_frame_errors, handoff_operators)
from ._ffi import LpvMpcError, STATUS_TEXT  # noqa: F401
from .track import Map  # noqa: F401

__all__ = ["BatchedSolver", "PathFollowingLPV_MPC", "LPV_MPC_Planner", "PlannerHandoff", "body_frame_errors", "handoff_operators",
           "Map", "LpvMpcError", "STATUS_TEXT"]
