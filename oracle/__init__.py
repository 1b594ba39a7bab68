"""CPU oracle for the KSD-gradient hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy (fp64) restatement of the reference algorithm
(sozoluffy/TensorNetworks) for the one path this repository accelerates:

    parameterised circuit -> Born probabilities q_theta -> Stein-kernel Gram K_p
    -> KSD = sqrt(q^T K_p q) -> parameter-shift gradient.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the reported CPU
baseline, never as the thing that is shipped.  Nothing under
``tensornetworks_amd/`` imports this package; the product path fails loudly
when the HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):

* Stein half (``oracle.stein``, ``oracle.ksd``): PINNED.  Checked against the
  reference's own known answers (stein_utils.py:205-251) and against golden
  vectors captured in this container by importing the reference's
  ``stein_utils`` / ``bayesian_network`` / ``utils`` / ``ksd_vi`` modules
  (``tests/golden/make_golden.py`` is the generating script).
* Circuit half (``oracle.circuit``): PARITY UNPINNED by the reference.  The
  reference simulates circuits with PennyLane ``default.qubit``
  (quantum_born_machine.py:4,:28,:58), a third-party dependency that is not
  vendored, not version-pinned (requirements.txt:1-4; README.md:72-75) and not
  installable here; the reference holds no test or fixture for q_theta.  The
  restatement follows PennyLane's published operator definitions and the gate
  order of quantum_born_machine.py:57-128, and is pinned only by analytic
  known-answer tests and by agreement of two independent implementations
  (gate-by-gate tensordot vs dense Kronecker unitaries).
"""
