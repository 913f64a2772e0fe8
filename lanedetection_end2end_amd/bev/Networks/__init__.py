"""Backbone registry -- same surface as BEV/Networks/__init__.py:9-20."""
from lanedetection_end2end_amd._refpath import extend as _extend
from .ERFNet import Net

model_dict = {'erfnet': Net}


def allowed_models():
    return model_dict.keys()


def define_model(mod, **kwargs):
    if mod not in allowed_models():
        raise KeyError("The requested model: {} is not implemented".format(mod))
    return model_dict[mod](**kwargs)


_extend(__path__, "Birds_Eye_View_Loss")
