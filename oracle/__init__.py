"""CPU oracle for the Smart-NINT ConvLSTM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (``nasa-niswan_amd``) never imports this package
and fails loudly when its HIP extension is missing.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the reference's own
``model.py`` (it needs only torch) in the build container, runs it on seeded
inputs and commits the input/output vectors under ``tests/golden/``;
``tests/test_oracle.py`` checks this restatement against those vectors, the
reference's parameter-count known-answer (test.ipynb:4698-4699) and the
notebook's 13x13 padding matrix (dataset_config.ipynb:484-502).
The arithmetic itself (conv / sigmoid / tanh / Adam) lives in PyTorch, which
the reference does not pin (README.md:23); the vectors were produced with
torch 2.10.0 CPU ops.
"""
from .convlstm_oracle import *  # noqa: F401,F403
from .preproc_oracle import *  # noqa: F401,F403
