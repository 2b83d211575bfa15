"""sam_road_amd — MI355X (gfx950) implementation of sam_road's tiled-inference hot path.

    from sam_road_amd import SAMRoad, load_config

`SAMRoad` is a drop-in for the reference's `model.SAMRoad` inference surface; the arithmetic lives in
hand-written HIP kernels behind the C ABI of include/samroad_hip.h (libsamroad_hip.so).
"""
from .config import Config, load_config  # noqa: F401
from .model import SAMRoad  # noqa: F401
from .tiling import get_patch_info_one_img  # noqa: F401
