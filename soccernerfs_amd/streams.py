"""Process-wide side streams by role.  HIP multiplexes a process's streams onto a few hardware queues (4 by default: tools/debug_queues.py), and chains that
share a queue do not overlap -- so the trainers of one process (bench.py builds three in a row) must not each create their own set: the ninth stream of the
process aliases whatever queue it lands on, and the overlap a trainer was designed around silently disappears (round 5: the config-4 leg's asynchronous sweep
ran at its alone speed, 2.06 ms, with no overlap when it followed the two K-Planes legs in one process, and overlapped as designed in a process of its own).
One stream per (device, role) for the whole process instead."""
from typing import Dict, Tuple

import torch

_POOL: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}

# (r06, measured and left out) A high-priority queue for the "adam" role -- the optimiser sweep / owner-computes tile pass is the critical chain of every trainer,
# what runs beside it is not -- changed nothing: config 4 4.03 / 4.01 ms per step with it, 3.99 / 4.00 without; K-Planes 1.808 / 1.820 M rays/s with it,
# 1.819 / 1.817 without.  What stretches those passes in the step is the memory system they share, not the order of workgroup dispatch.


def side_stream(device, role: str) -> "torch.cuda.Stream":
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    st = _POOL.get(key)
    if st is None:
        st = _POOL[key] = torch.cuda.Stream(device=dev)
    return st
