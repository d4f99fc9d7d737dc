"""Drop-in for the reference's ``lib/model/mpnn/__init__.py:1-7``: the same eight names."""
from fgnn_amd.mpnn import *  # noqa: F401,F403
from fgnn_amd.mpnn import (FactorNN, base_mp_nn, factor_mpnn, global_pooling, mp_conv_residual,  # noqa: F401
                           mp_conv_type, mp_conv_v2, mp_ensemble, mp_sequential)
