"""CPU: the C-ABI library builds/loads and exports every symbol include/mmidet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    txt = open(os.path.join(REPO, 'include', 'mmidet_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mmi_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from mmidet_hip import lib
    names = _declared()
    assert len(names) >= 20
    dll = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    assert lib._lib.mmi_version() >= 100
    # every declared symbol has a ctypes signature on the Python side, and vice versa
    assert sorted(lib.EXPORTS) == names


def test_argument_validation_without_gpu():
    """Bad descriptors are rejected on the host before any launch."""
    from mmidet_hip import lib
    d = lib.ConvDesc(1, 8, 8, 4, 8, 8, 4, 5, 5, 1, 2, 4, 4)   # 5x5 unsupported
    assert lib._lib.mmi_conv_fwd_row_blocks(ctypes.byref(d)) < 0
    assert b'unsupported' in lib._lib.mmi_last_error()
