"""tensornetworks_amd -- MI355X-native KSD variational inference with a quantum Born machine.

Drop-in for the hot path of sozoluffy/TensorNetworks (QuantumBornMachine, KSDVariationalInference,
stein_utils) on a hand-written HIP backend (libbornvi_hip.so, C ABI in include/bornvi.h).
"""
from .utils import generate_all_binary_outcomes, calculate_tvd  # noqa: F401

__all__ = ["QuantumBornMachine", "KSDVariationalInference", "generate_all_binary_outcomes", "calculate_tvd"]


def __getattr__(name):
    if name == "QuantumBornMachine":
        from .quantum_born_machine import QuantumBornMachine
        return QuantumBornMachine
    if name == "KSDVariationalInference":
        from .ksd_vi_quantum import KSDVariationalInference
        return KSDVariationalInference
    raise AttributeError(name)
