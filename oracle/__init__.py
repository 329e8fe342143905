"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the DPFT hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``dpft_amd``) never imports
this package and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * decoder / embeddings / querent / reference points / heads / loss:
    pinned against the reference's own Python, imported in the build container
    by ``oracle/gen_golden.py`` (fixtures in ``tests/golden``).
  * torchvision ResNet/FPN internals, the MSDA CUDA extension core and
    pytorch3d ``box3d_overlap`` are third-party code that is NOT present under
    /root/reference and the reference ships no tests for them:
    **parity unpinned** by the reference for those three cores; they are
    restated from their published algorithms on torch primitives.
"""
