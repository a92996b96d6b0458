"""CPU oracle for the aur_ppo GAE -> shuffle -> gather -> PPO-loss hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``aur_ppo_amd/`` may import this
package.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the
reported CPU baseline, never as the thing measured or shipped.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference
(``/root/reference/src/ppo.py`` under in-memory alias modules) in the build
container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors, and the MT19937 shuffle
additionally against ``numpy.random.RandomState`` itself (numpy is the
third-party dependency the reference calls, ``src/ppo.py:182,217``).
"""
