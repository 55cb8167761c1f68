"""TEST INFRASTRUCTURE ONLY — ctypes binding of the CPU oracle (oracle/picha_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (picha_amd) never does.
"""
from .binding import (  # noqa: F401
    OracleImage,
    build,
    color_convert,
    color_settings,
    lib,
    make_contribs,
    resize,
    resize_batch_mt,
)
