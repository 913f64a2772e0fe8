"""Backbone registry of this tree -- the surface of BP/Networks/__init__.py:9-20, bound to the shared table."""
from lanedetection_end2end_amd._refpath import extend as _extend
from lanedetection_end2end_amd.registry import make_registry
from .ERFNet import Net

model_dict, allowed_models, define_model = make_registry(erfnet=Net)

_extend(__path__, "Backprojection_Loss")
