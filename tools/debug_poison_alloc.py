#!/usr/bin/env python3
"""Run pytest targets with the caching allocator's free blocks pre-filled with a poison value: a kernel or module that reads a
torch.empty buffer before writing it sees the poison instead of the zeros a fresh hipMalloc often hands out.

    python tools/debug_poison_alloc.py nan tests/test_gpu_nerfplayer_full_trainer.py -k fifty
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
val = float(sys.argv[1])
blocks = [torch.full((mb * 262144,), val, device="cuda:0") for mb in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048] * 3]
small = [torch.full((n,), val, device="cuda:0") for n in [16, 64, 256, 1024, 4096, 16384, 65536] * 50]
del blocks, small
sys.exit(pytest.main(["-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + sys.argv[2:]))
