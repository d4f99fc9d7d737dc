"""fgnn_amd — MI355X-native FGNN message passing (host side of libfgnn_hip.so).

The public surface mirrors the reference's ``lib.model.mpnn`` package
(/root/reference/lib/model/mpnn/__init__.py:1-7); see INTEGRATION.md.
"""
from . import _hip, ops  # noqa: F401
from .mpnn import (FactorNN, base_mp_nn, factor_mpnn, mp_conv_residual, mp_conv_type,  # noqa: F401
                   mp_conv_v2, mp_sequential)
from .ldpc import LDPCModel  # noqa: F401
from .fastpath import disable_fast_path, enable_fast_path, fast_path  # noqa: F401  (FGNN_FAST_PATH=1 enables it at import)

__all__ = ['mp_conv_v2', 'mp_conv_type', 'mp_conv_residual', 'mp_sequential', 'factor_mpnn',
           'FactorNN', 'base_mp_nn', 'LDPCModel', 'enable_fast_path', 'disable_fast_path', 'fast_path']
