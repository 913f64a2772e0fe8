"""Backbone registry -- same surface as BP/Networks/__init__.py."""
from lanedetection_end2end_amd._refpath import extend as _extend
from .ERFNet import Net

model_dict = {'erfnet': Net}


def allowed_models():
    return model_dict.keys()


def define_model(mod, **kwargs):
    if mod not in allowed_models():
        raise KeyError("The requested model: {} is not implemented".format(mod))
    return model_dict[mod](**kwargs)


_extend(__path__, "Backprojection_Loss")
