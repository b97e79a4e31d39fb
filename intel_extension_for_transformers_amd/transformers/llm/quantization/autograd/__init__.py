from .functions import matmul_kbit, qbits_acquire_type, qbits_woq_linear_ref_impl  # noqa: F401
