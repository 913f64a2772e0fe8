"""``Networks.ERFNet`` of the BEV tree: the HIP-backed backbone (see lanedetection_end2end_amd/erfnet.py)."""
from lanedetection_end2end_amd.erfnet import Decoder, DownsamplerBlock, Encoder, Net, UpsamplerBlock, non_bottleneck_1d  # noqa: F401
