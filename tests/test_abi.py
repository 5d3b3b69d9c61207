"""CPU-only checks of the boundary: the library loads, exports every symbol
include/fc_hip.h declares, and refuses to compute without a device."""

import os
import re

import numpy as np
import pytest

from firecode_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fc_hip.h but not exported"
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS)
    assert lib.fc_abi_version() == 1


def test_no_cpu_fallback_without_device():
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    import firecode_amd as fc

    with pytest.raises(fc.FirecodeHipDeviceError):
        fc.init(0)
    with pytest.raises(fc.FirecodeHipDeviceError):
        fc.rmsd.rmsd_and_max(np.zeros((4, 3)), np.ones((4, 3)))
    with pytest.raises(fc.FirecodeHipDeviceError):
        fc.pruner.prune_by_rmsd(np.zeros((3, 4, 3)), np.array(["C"] * 4), 0.5)


def test_argument_validation_precedes_device_use():
    import firecode_amd as fc

    with pytest.raises(fc.FirecodeHipInputError):
        fc.pruner.prune_by_rmsd(np.zeros((3, 4, 2)), np.array(["C"] * 4), 0.5)
    with pytest.raises(fc.FirecodeHipInputError):
        fc.embeds.rototranslate(np.zeros((2, 4, 3)), np.zeros((3, 3, 3)), np.zeros((2, 3)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "firecode_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"
                assert "cpu_ref" not in src


def test_rccl_loader_survives_a_name_that_does_not_load():
    """fc_comm.cpp's loader tries FC_RCCL_LIB first and falls through to the standard names; a name that fails
    to dlopen must cost an error string, not the process (dlerror() may be read once per failure).  Without a
    device the call then ends in the library's ordinary no-device error."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r)\n"
            "from firecode_amd import _lib\n"
            "try:\n"
            "    _lib.comm_unique_id(); print('ID')\n"
            "except _lib.FirecodeHipError as e:\n"
            "    print('ERR', type(e).__name__)\n") % ROOT
    env = dict(os.environ, FC_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout.split()[0] in ("ID", "ERR")
