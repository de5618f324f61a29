"""CPU suite: the C-ABI shared library loads and exports every symbol include/allocnet_amd.h
declares; the ctypes table covers the header; without a GPU the product path fails loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "allocnet_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(anet_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from allocnet_amd import _lib
    names = _declared()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    assert _lib.load().anet_abi_version() == 2


def test_no_cpu_fallback():
    """On a box without a GPU creating a context must fail with ANET_ERR_NODEVICE (never a silent
    CPU path); on a GPU box it must succeed."""
    import allocnet_amd as aa
    from allocnet_amd import _lib
    lib = _lib.load()
    if lib.anet_device_count() == 0:
        with pytest.raises(aa.AnetError) as ei:
            aa.Context(0)
        assert ei.value.code == _lib.ANET_ERR_NODEVICE
        with pytest.raises(aa.AnetError):
            aa.minco_solve(np.zeros((1, 3, 3)), np.zeros((1, 3, 3)), np.zeros((1, 0, 3)), np.ones((1, 1)), 4)
    else:
        aa.Context(0).close()


def test_oracle_is_not_imported_by_the_product():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "allocnet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("test oracle", ""), f"{f} mentions the oracle"
    for dp, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dp, f)).read()
    # the measurement scripts under tools/ do not import it either (bench.py's cpu_baseline leg is the one exception)
    import re
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f"tools/{f} imports the oracle"


def test_cpp_facade_compiles_as_cxx14():
    """The facade headers must build with the reference's language level (-std=c++14,
    src/planner/CMakeLists.txt:4) and without Eigen or HIP headers on the include path."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "test_facade.cpp")
    res = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                          "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
