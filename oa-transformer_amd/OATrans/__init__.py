"""OATrans on MI355X.  The package mirrors the reference's module paths."""
import os as _os

# RCCL over xGMI inside one node needs dmabuf IPC on this driver stack; the variable is read when the HIP runtime
# initialises (first GPU call), which is after this import in every entry point.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
