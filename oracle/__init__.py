"""CPU oracle for the RegT-GCN forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and there only as the checker / the timed CPU baseline.  The product
package (``regt-gcn_amd/``) never imports this package and raises if its HIP library
is missing.

What is restated here (reference = raynbowy23/RegT-GCN, paths relative to the
reference root):

* ``oracle.graph_ops``  -- the two third-party graph operators the reference calls,
  ``torch_geometric.nn.GCNConv`` and ``torch_geometric.nn.ChebConv(K=2)``
  (call sites models/utils.py:107-113,169,175,181; models/RegionalTemporalGCN.py:77-80,
  136-140; models/TemporalGCN.py:65-69,88).  torch_geometric is an un-vendored,
  unpinned pip dependency (README.md:31) that is absent from this image, so the
  arithmetic is restated from the published algorithm (Kipf & Welling GCN
  normalisation; Defferrard Chebyshev filter with PyG's ``get_laplacian`` /
  ``lambda_max = 2 * max(w)`` conventions).
  **Parity status of these two operators: UNPINNED** -- the reference holds no test,
  fixture or golden vector for them; they are pinned here only against dense-matrix
  known answers (tests/test_oracle_ops.py).
* ``oracle.model``      -- the reference's own orchestration (RegionalA3TGCN,
  RegionalTemporalGCN, A3TGCN, TemporalGCN, TGCN cell), op for op in the reference's
  order.  **Parity status: PINNED** by golden vectors produced in the build container by
  importing the reference's model files themselves (oracle/make_goldens.py) with the
  two graph operators above injected for the missing torch_geometric package, including
  one run on the reference's shipped checkpoint
  ``pretrained/occrate/RegionalTemporalGCN/model_in6_out1_epoch50.pt``.
* ``oracle.loop``       -- the train()/test() loop semantics of run.py:163-226.
"""
