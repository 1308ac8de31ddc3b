"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (palettenerf_amd) never imports it; it fails loudly without its HIP library.
"""
from .orc import *  # noqa: F401,F403
