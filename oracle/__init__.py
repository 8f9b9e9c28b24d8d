"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A from-scratch CPU restatement (torch-CPU fp32 + numpy) of the reference's
``FBDDPGAgent.update()`` hot path (facebookresearch/controllable_agent,
``url_benchmark/agent/fb_ddpg.py:427-520``) used as the *checker* for the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package
``controllable_agent_amd`` never does (tests/test_no_oracle_in_product.py
enforces that).

Parity status: PINNED against the reference itself, imported in the dev
container (tests/golden/make_golden.py generated tests/golden/*.npz|json from
the real ``FBDDPGAgent``/``ReplayBuffer``; tests/test_oracle_golden.py replays
them through this restatement).  The reference's own test-suite holds no golden
vector for this path.
"""
