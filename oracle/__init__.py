"""ORACLE - CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: no TensorFlow in the build container and no golden vectors in the reference (see
keras_ref.py).  Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package never imports it.
"""
