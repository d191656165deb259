"""`flash_helpers` distribution -- counterpart of the reference's py/setup.py:6-9: kernel-config
enumeration, the test entry and the test/benchmark utilities under their reference import names.
The modules are aliases of flash_attention_from_scratch_amd.flash_helpers (installed by the root
setup.py), so this distribution depends on `flash_attention`.  The alias package lives INSIDE this
project (py/flash_helpers), so sdists and isolated builds contain it; the repository root's
flash_helpers/__init__.py only points in-tree imports at it."""
from setuptools import setup

__version__ = "0.3.0"

setup(
    name="flash_helpers",
    version=__version__,
    packages=["flash_helpers", "flash_helpers.test"],
    install_requires=["flash_attention"],
)
