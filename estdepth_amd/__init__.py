"""estdepth_amd -- MI355X (gfx950) native implementation of ESTDepth's plane-sweep + EST-transformer
hot path behind the reference's Python operator API.

    from estdepth_amd import DepthNetHybrid            # hybrid_models/model_hybrid.py
    from estdepth_amd import DepthHybridDecoder        # hybrid_models/hybrid_depth_decoder.py
    from estdepth_amd import EpipolarTransformer       # transformer/epipolar_transformer.py
    from estdepth_amd import homo_warping, warp_volume # utils/homo_utils.py

The arithmetic of the hot path lives in lib/libestd_hip.so (C ABI: include/estd_hip.h); there is no
CPU or eager fallback.
"""
from .model_hybrid import DepthNetHybrid, abs_rel  # noqa: F401
from .hybrid_depth_decoder import DepthHybridDecoder, depthlayer  # noqa: F401
from .epipolar_transformer import EpipolarTransformer  # noqa: F401
from .homo_utils import homo_warping, warp_volume, set_id_grid  # noqa: F401
from .layers_op import convbn, convbnrelu, convbn_3d, convbnrelu_3d, convbntanh_3d  # noqa: F401

__version__ = "0.1.0"
