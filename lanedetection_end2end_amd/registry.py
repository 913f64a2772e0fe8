"""Backbone registry shared by the two mirrored trees.

The reference keeps one per tree (``Networks.model_dict`` / ``allowed_models()`` / ``define_model(mod, **kwargs)``,
BEV/Networks/__init__.py:9-20, called from LSQ_layer.py:244-247 with ``--mod``); both trees here bind the same three names
to one table built by ``make_registry``.  An unknown name raises ``KeyError`` with the reference's message.
"""


def make_registry(**backbones):
    """-> (model_dict, allowed_models, define_model) over the given name -> constructor table."""
    table = dict(backbones)

    def allowed_models():
        return table.keys()

    def define_model(mod, **kwargs):
        ctor = table.get(mod)
        if ctor is None:
            raise KeyError("The requested model: {} is not implemented".format(mod))
        return ctor(**kwargs)

    return table, allowed_models, define_model
