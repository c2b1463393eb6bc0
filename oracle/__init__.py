"""CPU oracle for the MAGIC hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  Nothing under
``magic_amd/`` imports it; the product path fails loudly without its HIP library.

What pins it
------------
* ``sce_loss``, the KNN detector (``model/eval.py``), ``utils/poolers.py`` and
  ``utils/utils.py`` of the reference import and run in the authoring container
  (with an inert ``dgl`` placeholder for ``model/eval.py``); golden vectors made
  from them are committed under ``tests/golden/`` together with the generating
  script ``tests/golden/make_golden.py``.  Those parts of the oracle are pinned.
* The GAT layer / auto-encoder arithmetic of the reference lives in DGL 1.0.0
  (``requirements.txt:5``), which is not installed and cannot be installed here.
  The restatement in ``gat_oracle.py`` follows the reference call sites line by
  line and DGL's documented operator semantics; for those functions
  **parity is unpinned** by any executable reference output.
"""
