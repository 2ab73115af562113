"""`Instances`: per-image field container returned by the detector (fields `pred_boxes`, `scores`,
`pred_classes`) and by the RPN (`proposal_boxes`, `objectness_logits`).

Behavioural contract restated from reference detectron2/structures/instances.py:7-185:
  * attribute access is field access (names not starting with "_"); unknown field -> AttributeError;
  * every field must have the same `len` as the fields already present (AssertionError otherwise);
  * indexing / masking applies to every field and returns a new Instances; an int index keeps length 1;
  * `len()` of an Instances without fields raises NotImplementedError; iteration is refused;
  * `Instances.cat` needs equal `image_size` and joins tensors (torch.cat), lists (+) and any type exposing a
    `cat` classmethod (e.g. Boxes); `.to()` forwards to every field that has `.to`.
"""
import torch


def _join(values):
    head = values[0]
    if torch.is_tensor(head):
        return torch.cat(values, dim=0)
    if isinstance(head, list):
        merged = []
        for v in values:
            merged.extend(v)
        return merged
    joiner = getattr(type(head), "cat", None)
    if joiner is None:
        raise ValueError("Unsupported type {} for concatenation".format(type(head)))
    return joiner(values)


class Instances:
    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for name, value in fields.items():
            self.set(name, value)

    # ---- field protocol -----------------------------------------------------------------------
    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, value):
        if name[:1] == "_":
            object.__setattr__(self, name, value)
            return
        self.set(name, value)

    def __getattr__(self, name):
        fields = self.__dict__.get("_fields")
        if fields is None or name not in fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return fields[name]

    def set(self, name, value):
        n = len(value)
        if self._fields:
            assert n == len(self), "Adding a field of length {} to a Instances of length {}".format(n, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def remove(self, name):
        self._fields.pop(name)

    def get_fields(self):
        return self._fields

    # ---- container protocol -------------------------------------------------------------------
    def __len__(self):
        if not self._fields:
            raise NotImplementedError("Empty Instances does not support __len__!")
        return len(next(iter(self._fields.values())))

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    def __getitem__(self, item):
        if type(item) is int:
            n = len(self)
            if not -n <= item < n:
                raise IndexError("Instances index out of range!")
            item = slice(item, None, n)  # keeps a leading dimension of 1
        return Instances(self._image_size, **{k: v[item] for k, v in self._fields.items()})

    def to(self, *args, **kwargs):
        moved = {k: (v.to(*args, **kwargs) if hasattr(v, "to") else v) for k, v in self._fields.items()}
        return Instances(self._image_size, **moved)

    @staticmethod
    def cat(instance_lists):
        assert len(instance_lists) > 0 and all(isinstance(i, Instances) for i in instance_lists)
        first = instance_lists[0]
        if len(instance_lists) == 1:
            return first
        assert all(i.image_size == first.image_size for i in instance_lists[1:])
        return Instances(first.image_size,
                         **{k: _join([i.get(k) for i in instance_lists]) for k in first.get_fields()})

    def __repr__(self):
        body = ", ".join("{}: {}".format(k, v) for k, v in self._fields.items())
        return "Instances(num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self), self._image_size[0], self._image_size[1], body)

    __str__ = __repr__
