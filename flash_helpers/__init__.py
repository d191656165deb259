"""In-tree import name of the `flash_helpers` distribution.  The package itself lives in py/flash_helpers (the reference
keeps it under py/ too: py/setup.py:6-9); this file only points Python at it, so the repository holds ONE copy."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "py", "flash_helpers")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _f
