import importlib

registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = (entry_point, kwargs.get("kwargs", {}))


def make(id, **kwargs):
    entry_point, default_kwargs = registry[id]
    if isinstance(entry_point, str):
        mod_name, attr = entry_point.split(":")
        entry_point = getattr(importlib.import_module(mod_name), attr)
    kw = dict(default_kwargs)
    kw.update(kwargs)
    return entry_point(**kw)
