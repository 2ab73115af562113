"""Hierarchical config node with yaml `_BASE_` inheritance.

Behavioural contract (what `configs/*.yaml` and the CLI rely on), restated from
the reference's use of fvcore/yacs `CfgNode` (un-vendored third-party;
call sites: reference `detectron2/config/config.py:11-65`,
`lvc/config/config.py:6-64`, `lvc/engine/defaults.py:139-144`):

* attribute access == item access; nested dicts become nodes;
* `merge_from_file` resolves `_BASE_` (relative to the including file) first,
  then overlays the file; unknown keys raise `KeyError`;
* `merge_from_list(["A.B", value, ...])` literal-evals strings and coerces
  tuple<->list (and int->float) to the type of the existing value, any other
  type change raises `ValueError`;
* `freeze()/defrost()/is_frozen()/clone()/dump()`.
"""
import copy
import os
from ast import literal_eval

import yaml

BASE_KEY = "_BASE_"
_VALID_TYPES = {tuple, list, str, int, float, bool, type(None)}


class CfgNode(dict):
    IMMUTABLE = "__immutable__"
    NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        key_list = [] if key_list is None else key_list
        converted = {}
        for k, v in init_dict.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = type(self)(v, key_list=key_list + [k])
            converted[k] = v
        super().__init__(converted)
        self.__dict__[CfgNode.IMMUTABLE] = False
        self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed

    # -- attribute protocol -------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(
                "Attempted to set {} to {}, but CfgNode is immutable".format(name, value)
            )
        if name in self.__dict__:
            raise AttributeError("Invalid attempt to modify internal CfgNode state: {}".format(name))
        if isinstance(value, dict) and not isinstance(value, CfgNode):
            value = type(self)(value)
        self[name] = value

    def __str__(self):
        return self.dump()

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, super().__repr__())

    # -- (de)serialisation --------------------------------------------------
    def to_dict(self):
        def conv(node):
            if isinstance(node, CfgNode):
                return {k: conv(v) for k, v in node.items()}
            if isinstance(node, tuple):
                return list(node)
            return node

        return conv(self)

    def dump(self, **kwargs):
        kwargs.setdefault("default_flow_style", None)
        return yaml.safe_dump(self.to_dict(), **kwargs)

    @staticmethod
    def load_yaml_with_base(filename, allow_unsafe=False):
        with open(filename, "r") as f:
            try:
                cfg = yaml.safe_load(f)
            except yaml.constructor.ConstructorError:
                if not allow_unsafe:
                    raise
                f.seek(0)
                cfg = yaml.unsafe_load(f)
        cfg = cfg or {}

        def overlay(a, b):
            # b <- a, recursively (a wins)
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    overlay(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base_file = cfg.pop(BASE_KEY)
            if base_file.startswith("~"):
                base_file = os.path.expanduser(base_file)
            if not (base_file.startswith("/") or "://" in base_file):
                base_file = os.path.join(os.path.dirname(filename), base_file)
            base = CfgNode.load_yaml_with_base(base_file, allow_unsafe=allow_unsafe)
            overlay(cfg, base)
            return base
        return cfg

    # -- merging --------------------------------------------------------------
    def merge_from_file(self, cfg_filename, allow_unsafe=True):
        loaded = self.load_yaml_with_base(cfg_filename, allow_unsafe=allow_unsafe)
        self.merge_from_other_cfg(type(self)(loaded))

    def merge_from_other_cfg(self, cfg_other):
        _merge_a_into_b(cfg_other, self, self, [])

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2 != 0:
            raise AssertionError(
                "Override list has odd length: {}; it must be a list of pairs".format(cfg_list)
            )
        if BASE_KEY in cfg_list[0::2]:
            raise AssertionError("The reserved key '{}' can only be used in files!".format(BASE_KEY))
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            parts = full_key.split(".")
            for sub in parts[:-1]:
                if sub not in d:
                    raise AssertionError("Non-existent key: {}".format(full_key))
                d = d[sub]
            sub = parts[-1]
            if sub not in d:
                raise AssertionError("Non-existent key: {}".format(full_key))
            value = _decode(v)
            value = _coerce(value, d[sub], sub, full_key)
            d[sub] = value

    # -- mutability -----------------------------------------------------------
    def freeze(self):
        self._immutable(True)

    def defrost(self):
        self._immutable(False)

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._immutable(flag)

    def clone(self):
        return copy.deepcopy(self)

    def is_new_allowed(self):
        return self.__dict__[CfgNode.NEW_ALLOWED]


def _decode(v):
    if isinstance(v, dict):
        return CfgNode(v)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(replacement, original, key, full_key):
    ot, rt = type(original), type(replacement)
    if rt == ot:
        return replacement
    if replacement is None or original is None:
        return replacement
    if isinstance(original, CfgNode) and isinstance(replacement, dict):
        return type(original)(replacement)
    for frm, to in ((tuple, list), (list, tuple), (int, float)):
        if rt == frm and ot == to:
            return to(replacement)
    raise ValueError(
        "Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
            ot, rt, original, replacement, full_key
        )
    )


def _merge_a_into_b(a, b, root, key_list):
    for k, v_ in a.items():
        full_key = ".".join(key_list + [k])
        v = _decode(copy.deepcopy(v_))
        if k in b:
            v = _coerce(v, b[k], k, full_key)
            if isinstance(v, CfgNode) and isinstance(b[k], CfgNode):
                _merge_a_into_b(v, b[k], root, key_list + [k])
            else:
                b[k] = v
        elif b.is_new_allowed():
            b[k] = v
        else:
            raise KeyError("Non-existent config key: {}".format(full_key))
