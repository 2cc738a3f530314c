"""Incremental decoding on the HIP model: the reference's generation/sampling.py surface."""
from .decoder import GraphDecoder                                                         # noqa: F401
from .id_space import IdSpace                                                              # noqa: F401
from .sampling import (add_interlacing_beam_marks, filling_sequence, get_batch, inverse_prompt_score,  # noqa: F401
                       magnify, shrink_beams, top_k_logits)
