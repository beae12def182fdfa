"""pyascore_amd -- MI355X-native implementation of pyAscore's ``PyAscore.score`` hot path.

Public surface mirrors ``pyascore`` for this path (pyascore/__init__.py:17): ``PyAscore``.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "PyAscore":
        from .ascore import PyAscore
        return PyAscore
    raise AttributeError(name)
