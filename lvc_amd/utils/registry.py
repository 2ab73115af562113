"""Name-keyed registry: the drop-in surface for META_ARCH / BACKBONE /
PROPOSAL_GENERATOR / ROI_HEADS plug-ins.

Semantics follow the reference's use of fvcore's `Registry` (third-party, reached
through reference `detectron2/utils/registry.py:4`): `register()` works as a
decorator or as a call, the key is `obj.__name__`, registering a duplicate name
asserts, `get(name)` of an unknown name raises `KeyError`.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, (
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        )
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:

            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class

            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def __repr__(self):
        return "Registry of {}: {}".format(self._name, sorted(self._obj_map))
