"""cogview_amd -- MI355X-native implementation of the CogView training / inference hot path.

Sub-packages mirror the reference's module surface so that its training and sampling scripts can drive this
package unchanged:  cogview_amd.mpu, cogview_amd.model, cogview_amd.fp16, cogview_amd.vqvae.
All arithmetic runs in libcogview_hip.so (hand-written HIP for gfx950, include/cogview_hip.h); there is no
CPU fallback."""
__version__ = "0.1.0"

# mpu first: functional <-> mpu.transformer import each other, and this is the order that resolves (any sub-package can
# then be the first thing a script imports)
from . import mpu  # noqa: E402,F401
from .bind import bind_reference_names  # noqa: E402,F401
