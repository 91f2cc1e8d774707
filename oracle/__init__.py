"""ORACLE -- test infrastructure only.

CPU restatement (plain PyTorch fp32, no torch_geometric) of the reference's GNN
message-passing hot path.  It exists to *check* the HIP path and to serve as the
reported CPU baseline; it is never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import anything from this package; ``neural_lam_amd`` never does (enforced by
tests/test_boundary.py::test_product_never_imports_oracle).

Parity status: PINNED against golden vectors produced by the reference's own
Python files executed from /root/reference (tests/golden/make_golden.py), with
one caveat stated in every fixture: torch_geometric 2.3.1 itself is not
installable here, so ``MessagePassing.propagate``/``scatter`` ran through a
minimal stand-in that restates PyG's published semantics.
"""
