"""not-gpu tier: the C-ABI library loads and exports every symbol include/bbp.h declares; no compute without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bbp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bbp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(bbp):
    names = _declared()
    assert len(names) >= 15
    L = ctypes.CDLL(bbp.lib_path)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(bbp.SIGNATURES) == names  # the Python binding covers exactly the header


def test_no_cpu_fallback(bbp):
    """Without a gfx950 device the engine refuses to start (and says so) instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bbp.BbpError) as e:
        bbp.Context(0)
    assert e.value.status == 5


def test_sizes(bbp):
    assert bbp.lib.bbp_proof_record_size(8) == 1121 + 32 * 12 == bbp.record_size(8)
    assert bbp.lib.bbp_entropy_size(8) == 32 * 12 + 32 == bbp.entropy_size(8)


def test_product_does_not_reference_oracle():
    """The shipped package must never import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "dusk_blindbidproof_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "from oracle" not in txt and "import oracle" not in txt and "bbp_oracle" not in txt, f
