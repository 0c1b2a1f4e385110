import importlib

registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = (entry_point, kwargs)


def make(id, **kwargs):
    entry_point, reg_kwargs = registry[id]
    mod_name, attr = entry_point.split(":")
    cls = getattr(importlib.import_module(mod_name), attr)
    kw = dict(reg_kwargs.get("kwargs", {}))
    kw.update(kwargs)
    return cls(**kw)
