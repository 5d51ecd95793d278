"""Library-GEMM selection for the shapes of this model on gfx950.

The plain projections of the model (set-abstraction MLP, prediction heads, weight gradients) are library GEMMs
issued through ``torch.mm / bmm / addmm``; which rocBLAS / hipBLASLt kernel serves a shape is the library's heuristic
choice.  PyTorch's TunableOp can override that choice per shape from a table of measured winners.  This module ships
such a table for the shapes of the CoDA configurations on MI355X (``tunableop_gfx950.csv``, produced by
``tools/make_tunableop_file.py`` from tuning runs of ``bench.py``) and switches TunableOp on in LOOK-UP mode: no
tuning at run time, shapes that are not in the table keep the library's default -- same libraries, same fp32
arithmetic, a different tile configuration (measured: 464 -> 476 scenes/s on the headline step).

``build_model`` calls ``enable_tuned_gemms()`` once; ``CODA_TUNED_GEMMS=0`` or an explicit
``PYTORCH_TUNABLEOP_ENABLED`` in the environment leave TunableOp alone.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_STATE = {"done": False, "on": False}


def table_validators(path=TABLE):
    """{name: value} of the table's ``Validator`` lines (library versions and architecture it was measured on)."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.strip().split(",")
            if len(parts) >= 3 and parts[0] == "Validator":
                out[parts[1]] = ",".join(parts[2:])
    return out


def enable_tuned_gemms(path=TABLE):
    """Switch TunableOp to look-up mode with the shipped table.  Returns True when the table is in use.  Idempotent;
    a no-op without a GPU, on another architecture, when the table was measured with other library versions
    (TunableOp refuses it), or when the caller manages TunableOp through the environment."""
    if _STATE["done"]:
        return _STATE["on"]
    _STATE["done"] = True
    if os.environ.get("CODA_TUNED_GEMMS", "1") == "0" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return False
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
    if not arch.startswith("gfx950"):
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    ok = bool(tunable.read_file(path))
    if not ok:  # other library versions: the table's solution indices mean nothing there
        tunable.enable(False)
    _STATE["on"] = ok
    return ok


def is_on():
    return _STATE["on"]
