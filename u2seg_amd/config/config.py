"""yacs-compatible config tree with the surface detectron2 exposes (detectron2/config/config.py:12-265):
``get_cfg()``, ``CfgNode.merge_from_file`` with ``_BASE_`` inheritance, ``merge_from_list``,
``freeze/defrost/clone/dump`` and the ``@configurable`` / ``from_config`` constructor protocol."""
import copy
import functools
import inspect
import os
from ast import literal_eval

import yaml

_VALID_TYPES = (tuple, list, str, int, float, bool, type(None))
BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # ---- attribute access ----
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__["_frozen"]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__["_frozen"]

    def _set_frozen(self, flag):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__["_frozen"] = self.__dict__["_frozen"]
        return out

    # ---- loading ----
    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return {k: CfgNode._decode(x) for k, x in v.items()}
        if isinstance(v, str):
            try:
                return literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @classmethod
    def load_yaml_with_base(cls, filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base_file = cfg.pop(BASE_KEY)
            if base_file.startswith("~"):
                base_file = os.path.expanduser(base_file)
            if not os.path.isabs(base_file):
                base_file = os.path.join(os.path.dirname(filename), base_file)
            base = cls.load_yaml_with_base(base_file)
            merge_a_into_b(cfg, base)
            return base
        return cfg

    def merge_from_file(self, cfg_filename):
        loaded = self._decode(self.load_yaml_with_base(cfg_filename))
        self._merge_dict(loaded, [])

    def merge_from_other_cfg(self, other):
        self._merge_dict(other, [])

    def _merge_dict(self, a, path):
        if self.is_frozen():
            raise AttributeError("cannot merge into a frozen CfgNode")
        for k, v in a.items():
            full = ".".join(path + [k])
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    if k in self and self[k] is not None:
                        raise KeyError("config key {} is not a node".format(full))
                    self[k] = CfgNode()
                self[k]._merge_dict(v, path + [k])
            else:
                if k not in self:
                    raise KeyError("Non-existent config key: {}".format(full))
                self[k] = _coerce(v, self[k], full)

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                if sub not in d:
                    raise KeyError("Non-existent config key: {}".format(full_key))
                d = d[sub]
            if keys[-1] not in d:
                raise KeyError("Non-existent config key: {}".format(full_key))
            v = self._decode(v)
            d[keys[-1]] = _coerce(v, d[keys[-1]], full_key)

    def dump(self, **kwargs):
        def to_plain(n):
            if isinstance(n, CfgNode):
                return {k: to_plain(v) for k, v in n.items()}
            if isinstance(n, tuple):
                return [to_plain(x) for x in n]
            return n

        return yaml.safe_dump(to_plain(self), **kwargs)


def _coerce(new, old, key):
    """yacs' type discipline: same type, or tuple<->list, or int->float; None matches anything."""
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, CfgNode) and isinstance(new, dict):
        return CfgNode(new)
    raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(new), key))


def get_cfg():
    from .defaults import default_config

    return default_config()


def configurable(init_func=None, *, from_config=None):
    """Lets ``Cls(cfg, *a)`` call ``Cls.from_config(cfg, *a)`` and forward the returned kwargs to __init__
    (detectron2/config/config.py:130-200)."""
    if init_func is not None:
        assert inspect.isfunction(init_func) and from_config is None and init_func.__name__ == "__init__"

        @functools.wraps(init_func)
        def wrapped(self, *args, **kwargs):
            try:
                from_config_func = type(self).from_config
            except AttributeError as e:
                raise AttributeError("Class with @configurable must have a 'from_config' classmethod.") from e
            if _called_with_cfg(*args, **kwargs):
                explicit = _get_args_from_config(from_config_func, *args, **kwargs)
                init_func(self, **explicit)
            else:
                init_func(self, *args, **kwargs)

        return wrapped

    def wrapper(orig_func):
        @functools.wraps(orig_func)
        def wrapped(*args, **kwargs):
            if _called_with_cfg(*args, **kwargs):
                explicit = _get_args_from_config(from_config, *args, **kwargs)
                return orig_func(**explicit)
            return orig_func(*args, **kwargs)

        wrapped.from_config = from_config
        return wrapped

    return wrapper


def _get_args_from_config(from_config_func, *args, **kwargs):
    sig = inspect.signature(from_config_func)
    if list(sig.parameters.keys())[0] != "cfg":
        raise TypeError("{}.from_config must take 'cfg' as the first argument!".format(from_config_func))
    support_var_arg = any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in sig.parameters.values())
    if support_var_arg:
        return from_config_func(*args, **kwargs)
    supported = set(sig.parameters.keys())
    extra = {k: kwargs.pop(k) for k in list(kwargs.keys()) if k not in supported}
    ret = from_config_func(*args, **kwargs)
    ret.update(extra)
    return ret


def _called_with_cfg(*args, **kwargs):
    if len(args) and isinstance(args[0], CfgNode):
        return True
    if isinstance(kwargs.pop("cfg", None), CfgNode):
        return True
    return False
