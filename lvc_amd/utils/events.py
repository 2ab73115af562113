"""Minimal EventStorage (reference detectron2/utils/events.py): the modules call
`get_event_storage().put_scalar(...)` in training mode.  Unlike the reference, no active storage is not an
error here -- scalars are then dropped."""
from collections import defaultdict
from contextlib import contextmanager

_CURRENT = []


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._latest = {}
        self._history = defaultdict(list)

    def put_scalar(self, name, value, smoothing_hint=True):
        value = float(value)
        self._latest[name] = value
        self._history[name].append((self.iter, value))

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint)

    def latest(self):
        return self._latest

    def history(self, name):
        return self._history[name]

    def step(self):
        self.iter += 1

    def __enter__(self):
        _CURRENT.append(self)
        return self

    def __exit__(self, *exc):
        assert _CURRENT[-1] is self
        _CURRENT.pop()


class _NullStorage:
    def put_scalar(self, *a, **k):
        pass

    put_scalars = put_scalar


def get_event_storage():
    return _CURRENT[-1] if _CURRENT else _NullStorage()
