"""``trace`` decorator with the reference's shape (src/offmark/common/__logging.py:6-16):
logs ``Entering <fn>()`` at DEBUG on the given module logger."""
import functools


def trace(module_logger):
    def decorator(fn):
        @functools.wraps(fn)
        def inner(*args, **kwargs):
            module_logger.debug(f"Entering {fn.__name__}()")
            return fn(*args, **kwargs)
        return inner
    return decorator
