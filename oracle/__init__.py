"""CPU oracle for the ml-qem hot path -- TEST INFRASTRUCTURE ONLY.

A pure-torch (CPU, fp64-capable) restatement of the arithmetic the reference delegates to
torch_geometric>=2.0.0 / torch-sparse>=0.6.16 (requirements.txt:1-2; neither is installable here) and of the
reference's own models and feature encoders.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker / timed baseline.  Nothing under
``ml-qem_amd/`` imports it.

Parity status: PINNED by the reference's committed artefacts:
  G1  Family-B forward with docs/tutorials/model/ising_init_from_qasm_no_readout/gnn1.pth on the 300 graphs of
      .../data/ising_init_from_qasm_no_readout/val_extra/step_0.pk -> mean L2 0.117838
      (docs/tutorials/h17_compare_over_steps.ipynb:513)           [tests/test_oracle_goldens.py]
  G2  MLP1 + encode_data with .../mlp1_smaller_2.pth on the same circuits -> 0.032910 (same cell)
  G3  L2 of the raw noisy values -> 0.027510
Family A (GCN/Cheb/SAGE, docs/tutorials/01_ngem.ipynb cell [9]) has no checkpoint or printed output in the
reference -- nothing of the reference's can pin those three layers.  They are pinned instead by checks that share no
code with this package (tests/test_oracle_family_a.py): hand-computed answers on 3-node graphs and an independent
dense-matrix formulation in numpy fp64 covering duplicate edges, pre-existing self-loops, isolated nodes and
zero-out-degree sources, following PyG's published definitions.
"""
