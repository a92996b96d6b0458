#!/usr/bin/env python
"""Drop-in for the reference's src/run_ppo.py (its README's ``run.py``): same flags, same defaults."""
from aur_ppo_amd.run_ppo import main

if __name__ == "__main__":
    main()
