"""coda_neurips2023_amd -- MI355X (gfx950) native forward/backward hot path of CoDA.

Only what the hot path needs lives here:

* ``csrc/``       hand-written HIP kernels + the C ABI (``libcoda_hip.so``,
                  declared in ``include/*.h`` at the repo root)
* ``pointnet2/``  host-side mirror of the reference's ``pointnet2`` package
                  (``_ext`` operator module, autograd functions, SA module)
* ``transformer.py`` / ``helpers.py`` / ``position_embedding.py`` /
  ``model_3detr.py`` / ``criterion.py``  host-side mirrors of the reference
  modules of the same names (``models/*.py``, ``criterion.py``)

There is no CPU fallback: every operator raises if ``libcoda_hip.so`` is
missing or if it is handed a non-GPU tensor (as the reference's ops do,
``_ext_src/src/ball_query.cpp:30-32``).
"""

__version__ = "0.1.0"
