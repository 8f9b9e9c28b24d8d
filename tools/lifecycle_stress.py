"""Round-3 reproduction of the release-build segfault: many short-lived agents in ONE process, each capturing
the pipelined (two-branch) multi-step graph, the single-step graph and the batch-1 graphs, then dropped.  Reports how many
contexts are alive when a new one captures (reference cycles delay ``__del__`` until a cyclic GC pass), so that a crash can be
read against resource pressure.  usage: python tools/lifecycle_stress.py N [gc_every] [pickle]"""
import gc
import sys
import weakref

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import fb_oracle as fo          # noqa: E402
from tests import helpers as H              # noqa: E402
from tests.test_update_parity_gpu import _buffer    # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
GC_EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 0
PICKLE = len(sys.argv) > 3 and sys.argv[3] == "pickle"
rng = np.random.default_rng(3)
cfg = fo.OracleConfig(obs_dim=5, action_dim=3, goal_dim=5, use_goal=False, z_dim=8, hidden_dim=32, feature_dim=16,
                      backward_hidden_dim=18, batch_size=16, lr=1e-3)
nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim, None)
rb = _buffer(storage, lengths, cfg.discount, cfg.future)
alive = []
if GC_EVERY == 0:
    gc.disable()
for i in range(N):
    a = H.make_hip_agent(cfg, nets, None)
    alive.append(weakref.ref(a))
    a.update(rb, 0)
    a.update_many(rb, 1, 3)
    a.update_many(rb, 4, 2)
    m = a.init_meta()
    a.act(rng.standard_normal(cfg.obs_dim).astype(np.float32), m, 0, eval_mode=True)
    if PICKLE:
        import pickle
        b = pickle.loads(pickle.dumps(a))
        b.update_many(rb, 6, 3)
        alive.append(weakref.ref(b))
        del b
    del a
    if GC_EVERY and (i + 1) % GC_EVERY == 0:
        gc.collect()
    if (i + 1) % 25 == 0:
        n_alive = sum(1 for r in alive if r() is not None)
        print(f"{i + 1} agents built, {n_alive} still alive, mem {torch.cuda.memory_allocated() >> 20} MiB", flush=True)
torch.cuda.synchronize()
print("stress ok", N, GC_EVERY, PICKLE, flush=True)
