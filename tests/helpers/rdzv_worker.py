"""Launched by tests/test_dist_gloo.py under `python -m torch.distributed.run`: the torch-free rendezvous of the
bench (lgmi.dist.group_from_env) next to torchrun's own agent store."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]

from lgmi.dist import exchange_unique_id, group_from_env  # noqa: E402

g = group_from_env(timeout=60.0)
uid = exchange_unique_id(g, (lambda: bytes(range(128))) if g.rank == 0 else None)
allr = g.allgather({'rank': g.rank, 'pid': os.getpid()})
g.barrier()
mx = g.allreduce_max(10.0 + g.rank)
got = g.gather(g.rank * 7)
with open(os.path.join(sys.argv[1], 'rank%d.json' % g.rank), 'w') as f:
    json.dump({'uid_ok': uid == bytes(range(128)), 'ranks': [a['rank'] for a in allr], 'max': mx, 'gathered': got,
               'torch_loaded': 'torch' in sys.modules, 'agent_store': os.environ.get('TORCHELASTIC_USE_AGENT_STORE')}, f)
g.close()
