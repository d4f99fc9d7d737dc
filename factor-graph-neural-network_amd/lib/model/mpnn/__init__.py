from fgnn_amd.mpnn import *  # noqa: F401,F403
from fgnn_amd.mpnn import (FactorNN, base_mp_nn, factor_mpnn, mp_conv_residual, mp_conv_type,  # noqa: F401
                           mp_conv_v2, mp_sequential)
