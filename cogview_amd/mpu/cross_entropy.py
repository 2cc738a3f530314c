"""vocab_parallel_cross_entropy (mpu/cross_entropy.py:107-109) -> fused HIP kernels."""
from ..functional import vocab_parallel_cross_entropy  # noqa: F401
