"""`flash_attention` distribution -- counterpart of the reference's root setup.py:45-75.

The reference builds a CUDAExtension named `flash_attention_kernels`; here the device code is
libfa_hip.so (hand-written HIP for gfx950 behind a C ABI, built by csrc/Makefile with hipcc -- it
cross-compiles without a GPU) and `flash_attention_kernels` is a pure-Python module over it (ctypes),
so no torch headers are compiled.  Installs:

    flash_attention                  the API package (forward, forward_timed, forward_ex)
    flash_attention_kernels          the extension module name the reference binds (module alias)
    flash_attention_from_scratch_amd the MI355X package: _capi, API mirror, flash_helpers, tools,
                                     and lib/libfa_hip.so as package data

`flash_helpers` installs from ./py (the reference's second package, py/setup.py:6-9).
"""
import os
import subprocess
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

__version__ = "0.3.0"
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = "flash_attention_from_scratch_amd"


class BuildWithLibrary(build_py):
    """build_py that first generates the variant list and builds libfa_hip.so in-tree (make + hipcc)."""

    def run(self):
        if os.environ.get("FA_SKIP_NATIVE_BUILD", "") != "1":
            subprocess.run([sys.executable, os.path.join(HERE, PKG, "tools", "generate_kernel_instantiations.py")],
                           check=True)
            jobs = str(max(2, min(8, os.cpu_count() or 2)))
            subprocess.run(["make", "-C", os.path.join(HERE, PKG, "csrc"), "-j", jobs], check=True)
        lib = os.path.join(HERE, PKG, "lib", "libfa_hip.so")
        if not os.path.exists(lib):
            raise RuntimeError(f"{lib} is missing: the package has no CPU fallback and is useless without it")
        super().run()


setup(
    name="flash_attention",
    version=__version__,
    description="Flash-Attention-2 forward for AMD MI355X (gfx950): hand-written HIP behind a C ABI",
    packages=["flash_attention"] + [p for p in find_packages(include=[PKG, PKG + ".*"])],
    py_modules=["flash_attention_kernels"],
    package_data={PKG: ["lib/libfa_hip.so", "csrc/*", "tools/*.hip", "tools/*.inc", "tools/*.sh"]},
    include_package_data=False,
    cmdclass={"build_py": BuildWithLibrary},
    install_requires=["torch", "einops"],
    python_requires=">=3.9",
)
