#!/usr/bin/env python
"""Drop-in for the reference's src/robot_run.py: same flags, same defaults."""
from aur_ppo_amd.robot_run import main

if __name__ == "__main__":
    main()
