"""Config base class: attribute bag with deep-copying ``override`` (interface of the reference's
``routeformer/utils/config.py:7-33``: ``cfg[item]``, ``.get``, ``.copy``, ``.override(**kw)``)."""
import copy
from argparse import Namespace


class BaseConfig(Namespace):
    def __getitem__(self, name):
        return getattr(self, name)

    def get(self, name, default):
        return getattr(self, name, default)

    def copy(self):
        return copy.deepcopy(self)

    __copy__ = copy

    def override(self, **changes):
        """Deep copy with ``changes`` applied; derived fields are recomputed via ``__post_init__``."""
        new = self.copy()
        for name, value in changes.items():
            setattr(new, name, value)
        post = getattr(new, "__post_init__", None)
        if post is not None:
            post()
        return new
