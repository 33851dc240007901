"""CPU oracle for the Chebyshev graph-convolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.  ``gcn_fmri_decoding_amd`` must not
import from here (``tests/test_abi_and_host.py::test_product_does_not_import_oracle``
enforces that).

What it is: a NumPy/SciPy restatement of the algorithm of
``zhangyu2ustc/GCN_fmri_decoding`` for the path named by BASELINE.json
(citations are relative to the reference checkout):

* ``graph_ref``      -- ``lib_new/graph.py:79-98`` (laplacian), ``:146-152``
                        (rescale_L), ``:155-172`` (chebyshev recurrence)
* ``coarsening_ref`` -- ``lib_new/coarsening.py`` (coarsen / metis /
                        metis_one_level / compute_perm / perm_data(_3d) /
                        perm_adjacency)
* ``layers_ref``     -- ``lib_new/models_gcn.py:587-682`` (chebyshev5, b1relu,
                        b2relu, mpool1, apool1, fc, _inference), ``:253-276``
                        (loss), ``:278-313`` (Adam step, TF form) with an
                        explicit hand-derived backward pass.
* ``loop_ref``       -- ``lib_new/models_gcn.py:31-184`` (predict / evaluate /
                        fit: the callers of the path) around ``layers_ref.Net``.
* ``torch_cpu_ref``  -- the same network on torch CPU tensors (all host cores);
                        exists only as the timed CPU baseline B2 of bench.py.

Parity pinning (DESIGN.md "Oracle"):
* ``graph_ref`` and ``coarsening_ref`` are pinned against the reference's own
  importable modules run in the build container (``oracle/gen_golden.py``
  -> ``tests/golden/*.npz``) and against the reference's only known-answer
  test (``lib_new/coarsening.py:217-218``).
* ``layers_ref`` forward is pinned against ``lib_new/models_gcn.py`` layer
  methods executed *verbatim* under a NumPy stand-in for the handful of
  TensorFlow symbols they touch (same script).  TensorFlow itself is absent
  from the image, so the primitive-op semantics inside that stand-in are our
  reading of TF-1 documentation: the TF boundary itself is "parity unpinned".
* ``layers_ref`` backward has no reference counterpart that can run here (TF
  autodiff); it is checked against ``torch.autograd`` in float64 and finite
  differences (``tests/test_oracle_layers.py``).
"""
