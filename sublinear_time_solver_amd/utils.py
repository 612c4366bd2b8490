"""solver::utils (src/solver/mod.rs:363-461) — the reference's helper module, vectors reduced on the device through the C ABI."""
from .solver import check_convergence, compute_norm, compute_residual, l1_norm, l2_norm, linf_norm  # noqa: F401

__all__ = ["l2_norm", "l1_norm", "linf_norm", "compute_norm", "compute_residual", "check_convergence"]
