"""CPU oracle for the Multi-StyleGAN G+D training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the reference algorithm; it exists so that the HIP path in
``multi_stylegan_amd`` can be checked against something independent of it.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- and only as the checker, never as the thing
shipped or measured.  Nothing under ``multi_stylegan_amd/`` imports it.

Pinning: every function here is checked against golden vectors produced by
importing the reference's own Python modules in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``); see DESIGN.md "Oracle".
Each function cites the reference file:line (relative to the reference repo
root) it restates.
"""
