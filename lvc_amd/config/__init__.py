"""Config surface of the hot path: `get_cfg()`, `CfgNode`, `global_cfg`, `set_global_cfg`.

Mirrors reference `lvc/config/config.py:6-95` (which layers on
`detectron2/config/config.py:11-65`): same key space and defaults (data in
`defaults.yaml`, see its header), versioned `merge_from_file` (only VERSION-2
files are accepted here: every shipped `configs/*.yaml` is v2; a v1 file raises),
and the mutable process-global `global_cfg` that `CascadeROIHeads.forward` reads
(reference `lvc/modeling/roi_heads/cascade_rcnn.py:145`).
"""
import os

import yaml

from .cfgnode import CfgNode as _CfgNode

_DEFAULTS_FILE = os.path.join(os.path.dirname(__file__), "defaults.yaml")
LATEST_VERSION = 2


class CfgNode(_CfgNode):
    def merge_from_file(self, cfg_filename, allow_unsafe=True):
        assert os.path.isfile(cfg_filename), "Config file '{}' does not exist!".format(cfg_filename)
        loaded = type(self)(self.load_yaml_with_base(cfg_filename, allow_unsafe=allow_unsafe))
        assert self.VERSION == LATEST_VERSION, (
            "CfgNode.merge_from_file is only allowed on a config object of latest version!"
        )
        loaded_ver = loaded.get("VERSION", None)
        if loaded_ver is None:
            # reference compat.guess_version: v1 files carry MODEL.RPN_HEAD.NAME or MODEL.WEIGHT
            loaded_ver = 1 if ("WEIGHT" in loaded.get("MODEL", {})) else LATEST_VERSION
        assert loaded_ver <= self.VERSION, "Cannot merge a v{} config into a v{} config.".format(
            loaded_ver, self.VERSION
        )
        if loaded_ver != self.VERSION:
            raise NotImplementedError(
                "v{} config files are not supported by lvc_amd (all shipped configs are v2)".format(loaded_ver)
            )
        self.merge_from_other_cfg(loaded)


def _load_defaults():
    with open(_DEFAULTS_FILE) as f:
        blob = yaml.safe_load(f)
    cfg = CfgNode(blob["cfg"])
    for dotted in blob["__tuples__"]:
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = tuple(node[parts[-1]])
    return cfg


_C = _load_defaults()
global_cfg = CfgNode()


def get_cfg():
    """A fresh copy of the defaults (reference `lvc/config/config.py:70-78`)."""
    return _C.clone()


def set_global_cfg(cfg):
    """Point the process-global config at `cfg` (reference `lvc/config/config.py:81-95`)."""
    global_cfg.clear()
    global_cfg.update(cfg)


__all__ = ["CfgNode", "get_cfg", "global_cfg", "set_global_cfg"]
