"""Same export list as the reference's lib/model/mpnn/__init__.py:1-7."""
from .message_op import base_mp_nn, mp_conv_type, mp_conv_v2  # noqa: F401
from .blocks import (iid_mapping, iid_mapping_bn, iid_mapping_in, mp_conv_residual,  # noqa: F401
                     max_pool_layer, flatten)
from .pointwise import BatchNormAct2d, NodeInstanceNorm, PointwiseConv2d  # noqa: F401
from .assemblies import FactorNN, factor_mpnn, mp_sequential  # noqa: F401
from .combinators import global_pooling, identity_module, mp_ensemble, parallel_net  # noqa: F401
