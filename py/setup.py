"""`flash_helpers` distribution -- counterpart of the reference's py/setup.py:6-9: kernel-config
enumeration, the test entry and the test/benchmark utilities under their reference import names.
The modules are aliases of flash_attention_from_scratch_amd.flash_helpers (installed by the root
setup.py), so this distribution depends on `flash_attention`."""
import os

from setuptools import setup

__version__ = "0.2.0"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir)

setup(
    name="flash_helpers",
    version=__version__,
    packages=["flash_helpers", "flash_helpers.test"],
    package_dir={"flash_helpers": os.path.relpath(os.path.join(ROOT, "flash_helpers"))},
    install_requires=["flash_attention"],
)
