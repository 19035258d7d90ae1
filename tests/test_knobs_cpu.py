"""The knob table of INTEGRATION.md is the code's: every OAT_* environment variable the product or bench.py reads is documented there,
nothing documented is unread, and the product keeps to its budget of 15 (round 6: 13).  CPU only."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOT_KNOBS = {"OAT_DEV", "OAT_LAUNCH", "OAT_MAX_LDS", "OAT_TIME_DISPATCH", "OAT_LN_BWD", "OAT_LIN_NONE", "OAT_LIN_GELU", "OAT_LIN_RELU_IN",
             "OAT_LIN_EXACT"}            # C macros / constants, not environment variables
INTERNAL = {"OAT_BENCH_INIT", "OAT_BENCH_SELF_LAUNCHED"}          # set BY bench.py's launcher for its ranks


def _names(paths):
    found = set()
    for p in paths:
        found |= set(re.findall(r"\bOAT_[A-Z][A-Z0-9_]*\b", open(p).read()))
    return found - NOT_KNOBS


def test_documented_knobs_are_the_knobs_the_code_reads():
    pkg = os.path.join(ROOT, "oa-transformer_amd")
    product = _names(glob.glob(os.path.join(pkg, "OATrans", "**", "*.py"), recursive=True) + glob.glob(os.path.join(pkg, "csrc", "*.h*")) + [os.path.join(ROOT, "include", "oatrans_hip.h")])
    bench = _names([os.path.join(ROOT, "bench.py")])
    doc = _names([os.path.join(ROOT, "INTEGRATION.md")])
    assert (product | bench) - INTERNAL <= doc, sorted((product | bench) - INTERNAL - doc)
    assert doc - INTERNAL <= product | bench, sorted(doc - INTERNAL - product - bench)
    assert len(product) <= 15, sorted(product)
    # the kernel library itself reads no environment variable and exports no setter (tests/test_abi_cpu.py checks the binary)
    for path in glob.glob(os.path.join(pkg, "csrc", "*.h*")):
        assert "getenv" not in open(path).read(), path
