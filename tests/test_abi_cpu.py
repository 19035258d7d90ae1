"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every
symbol include/oatrans_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import subprocess

import pytest


def test_library_exports_every_declared_symbol():
    from OATrans.ops import hip
    if not os.path.exists(hip.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(hip.LIB_PATH), "csrc"), "-j8"], check=True)
    names = hip.declared_symbols()
    assert len(names) >= 20 and "oat_gemm_nt" in names and "oat_attn_space_bwd" in names
    lib = ctypes.CDLL(hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.oat_abi_version() == 2


def test_argument_validation_without_gpu():
    """Host-side argument checks run before any launch, so they are testable on CPU."""
    from OATrans.ops import hip
    lib = hip.lib()
    lib.oat_last_error.restype = ctypes.c_char_p
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    rc = lib.oat_gemm_nt(one, one, 128, 128, 100, 128, 128, 0, one, 128, null, 0, null, null, 0, 0, null, 0, 0, 0, null)
    assert rc < 0 and b"multiple of 64" in lib.oat_last_error()
    rc = lib.oat_gemm_nt(one, one, 0, 128, 128, 128, 128, 0, one, 128, null, 0, null, null, 0, 0, null, 0, 0, 0, null)
    assert rc < 0 and b"empty" in lib.oat_last_error()
    # launch policy is per call (tune / grid arguments): an unknown kernel choice or an out-of-range grid is refused before any launch
    rc = lib.oat_gemm_nt(one, one, 128, 128, 128, 128, 128, 0, one, 128, null, 0, null, null, 0, 0, null, 0, 3, 0, null)
    assert rc < 0 and b"kernel choice" in lib.oat_last_error()
    rc = lib.oat_gemm_nt(one, one, 128, 128, 128, 128, 128, 0, one, 128, null, 0, null, null, 0, 0, null, 0, 0, 0x10000, null)
    assert rc < 0 and b"grid" in lib.oat_last_error()
    rc = lib.oat_attn_space_fwd(one, 0, one, 0, one, 1, 1, 4, 2, 100, ctypes.c_float(0.125), null)
    assert rc < 0 and b"head_dim" in lib.oat_last_error()
    rc = lib.oat_attn_time_fwd(one, 0, one, 0, one, 1, 9, 4, 2, 128, ctypes.c_float(0.125), null)
    assert rc < 0 and b"frame counts" in lib.oat_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from OATrans.ops import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.OatError):
        hip.lib()


def test_the_library_keeps_no_tuning_state():
    """Round 6: no oat_*_set_* entry point is declared or exported, and no kernel source reads the environment - what a launch does is
    fixed by its arguments (include/oatrans_hip.h, Conventions)."""
    import glob
    from OATrans.ops import hip
    assert not [n for n in hip.declared_symbols() if "_set_" in n]
    lib = ctypes.CDLL(hip.LIB_PATH)
    for gone in ("oat_gemm_set_variant", "oat_gemm_set_m224", "oat_gemm_set_band", "oat_gemm_set_tail_split", "oat_gemm_set_tile_counters",
                 "oat_gemm_set_splitk_workspace", "oat_gemm_tn_set_variant", "oat_attn_time_set_variant", "oat_attn_space_set_variant"):
        assert not hasattr(lib, gone), gone
    csrc = os.path.join(os.path.dirname(hip.LIB_PATH), "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        assert "getenv" not in open(path).read(), path
