"""``trace(logger)`` decorator with the reference's behaviour (src/offmark/common/__logging.py:6-16):
each call of the wrapped function logs ``Entering <name>()`` at DEBUG on the given module logger."""
import functools
import logging


class trace:
    def __init__(self, module_logger):
        self._log = module_logger

    def __call__(self, fn):
        name = getattr(fn, "__name__", repr(fn))

        @functools.wraps(fn)
        def traced(*args, **kwargs):
            if self._log.isEnabledFor(logging.DEBUG):
                self._log.debug("Entering %s()", name)
            return fn(*args, **kwargs)

        return traced
