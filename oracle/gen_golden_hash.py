"""G9b: the reference's own pure-torch hash (HashEncoding.hash_fn, NS/field_components/encodings.py:289-307) on random integer grid
corners -- a second, importable implementation of the Instant-NGP spatial hash that pins the prime constants and the XOR/modulo
structure of the oracle's restatement of temporal_gridencoder.cu:46-59 (`tgrid_oracle.fast_hash`).  hash_fn multiplies in int64
without the uint32 wrap of the CUDA code; for power-of-two table sizes <= 2^32 the low bits -- all that survives the modulo -- are
identical.  Run in the build container only:  python oracle/gen_golden_hash.py

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle._refimport import import_reference  # noqa: E402

import_reference()
from nerfstudio.field_components.encodings import HashEncoding  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9b_hash.npz")
gen = torch.Generator().manual_seed(99)
res = {}
for log2 in (8, 17, 19):
    # num_levels=1 divides by zero in the growth factor: build with 2 levels and read level 0 (hash_offset[0] == 0)
    enc = HashEncoding(num_levels=2, min_res=16, max_res=32, log2_hashmap_size=log2, implementation="torch")
    pos = torch.randint(0, 2049, (512, 1, 3), generator=gen).expand(512, 2, 3).contiguous()
    h = enc.hash_fn(pos)[:, 0]
    res[f"pos_{log2}"] = pos[:, 0].numpy()
    res[f"hash_{log2}"] = h.numpy()
np.savez_compressed(out, **res)
print("wrote", out, os.path.getsize(out), "bytes")
