#!/usr/bin/env python3
"""LPV-MPC control node on the MI355X drop-in classes.  Put this file (and plannerMain.py) in place of the reference's
workspace/src/barc/src/controllerMain.py so that MAIN_LAUNCH.launch:114-118 (`type="controllerMain.py"`) starts it; the
repository root must be on PYTHONPATH and liblpvmpc.so built (`python __graft_entry__.py`)."""
from lpvmpc.ros_nodes import controller_main

if __name__ == "__main__":
    controller_main()
