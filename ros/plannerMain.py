#!/usr/bin/env python3
"""LPV-MPP trajectory-planner node on the MI355X drop-in classes (MAIN_LAUNCH.launch:38-47, `type="plannerMain.py"`);
see controllerMain.py next to this file."""
from lpvmpc.ros_nodes import planner_main

if __name__ == "__main__":
    planner_main()
