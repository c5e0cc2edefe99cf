"""TEST-ONLY stand-in for the `omegaconf` package (absent from this image and its wheelhouse), covering exactly the calls
wildgaussians/method.py makes: OmegaConf.structured / merge / load / from_dotlist / create / to_yaml / save
(method.py:447-455, 1649-1670, 2037).  A config is a plain attribute bag; values from YAML / dot-lists are coerced to the
dataclass field's declared type the way a structured config would."""
import dataclasses
import typing

import yaml


class _Cfg:
    def __init__(self, values=None, types=None):
        object.__setattr__(self, "_values", dict(values or {}))
        object.__setattr__(self, "_types", dict(types or {}))

    def __getattr__(self, k):
        try:
            v = self._values[k]
        except KeyError:
            raise AttributeError(k) from None
        if isinstance(v, str) and v == "???":
            raise AttributeError(f"Missing mandatory value: {k}")
        return v

    def __setattr__(self, k, v):
        self._values[k] = _coerce(v, self._types.get(k))

    __getitem__ = __getattr__

    def keys(self):
        return self._values.keys()

    def items(self):
        return self._values.items()

    def __contains__(self, k):
        return k in self._values


def _coerce(v, tp):
    if tp is None or v is None:
        return v
    origin = typing.get_origin(tp)
    if origin is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        return _coerce(v, args[0]) if len(args) == 1 else v
    if tp is bool:
        return v if isinstance(v, bool) else str(v).lower() in ("1", "true", "yes")
    if tp in (int, float, str):
        return tp(v)
    return v


class OmegaConf:
    @staticmethod
    def structured(cls):
        vals, types = {}, {}
        hints = typing.get_type_hints(cls)
        for f in dataclasses.fields(cls):
            types[f.name] = hints.get(f.name)
            if f.default is not dataclasses.MISSING:
                vals[f.name] = f.default
            elif f.default_factory is not dataclasses.MISSING:
                vals[f.name] = f.default_factory()
            else:
                vals[f.name] = "???"
        return _Cfg(vals, types)

    @staticmethod
    def create(obj=None):
        if isinstance(obj, str):
            obj = yaml.safe_load(obj)
        return _Cfg(obj or {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return _Cfg(yaml.safe_load(f) or {})

    @staticmethod
    def from_dotlist(items):
        out = {}
        for it in items:
            k, v = it.split("=", 1)
            out[k] = yaml.safe_load(v)
        return _Cfg(out)

    @staticmethod
    def merge(*cfgs):
        out = _Cfg(cfgs[0]._values, cfgs[0]._types)
        for c in cfgs[1:]:
            for k, v in c.items():
                if out._types and k not in out._types:
                    raise KeyError(f"Key '{k}' not in the structured config")
                setattr(out, k, v)
        return out

    @staticmethod
    def to_yaml(cfg, resolve=True):
        return yaml.safe_dump(dict(cfg._values))

    @staticmethod
    def save(cfg, path):
        with open(path, "w") as f:
            f.write(OmegaConf.to_yaml(cfg))
