"""limitador_amd — MI355X-native counter engine for Limitador's in-memory check_and_update path.

The package is a thin host layer over ``lib/librl_engine.so`` (hand-written gfx950 kernels behind
the C ABI of ``include/rl_engine.h``).  There is no CPU implementation behind it: importing
:mod:`limitador_amd.engine` without the built library raises, and creating an engine without a
MI355X raises ``EngineError(RL_ERR_NO_DEVICE)``.
"""
from .wire import HIT_DTYPE, CELL_ROW_DTYPE, LIMIT_ROW_DTYPE, RL_SIMPLE, make_hits  # noqa: F401

__all__ = ["HIT_DTYPE", "CELL_ROW_DTYPE", "LIMIT_ROW_DTYPE", "RL_SIMPLE", "make_hits"]
