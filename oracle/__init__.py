"""CPU oracle for the LPV-MPC / LPV-MPP solve path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import, link or execute it, and there only as the
checker, never as the thing measured.  The product path
(``autonomous-racing-lpv-mpp-mpc_amd/``) must never import from here.

Contents
--------
``lpv_ref.py``   numpy float64 restatement of the reference's LPV evaluation,
                 roll-out and dense QP assembly (pinned against golden vectors
                 captured from the reference itself, see ``tests/golden``).
``osqp_ref.c``   C float64 restatement of the OSQP ADMM algorithm the reference
                 calls through the third-party ``osqp`` wheel (NOT vendored in
                 the reference, version NOT pinned there => *parity unpinned*
                 for that stage; see the file header).
``osqp_ref.py``  ctypes front end for ``osqp_ref.c``.
``kkt_cert.py``  solver-independent KKT optimality certificate.
"""
