"""cogview_amd.generation on the HIP kernels, held to what the REFERENCE's generation code produced on the reference's own fp32
model (tests/golden/generate_samples.npz from oracle/gen_golden_generate.py: generation/sampling.py filling_sequence with two
beams and top_k = 1, inverse_prompt_score on two 1037-token rows): the same 40 image codes on both beams, the same two scores to
5e-3 -- with the reference's layer-input memories and with this package's in-place key/value cache.  The weights are drawn by
the constructor under the golden's seed (the reference's bits), nothing but token rows is stored."""
import pytest
import torch

from tests.generation_cases import run_generation_golden_case, run_sparse_generation_golden_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kv_cache", [False, True])
def test_generation_reproduces_the_reference_tokens_and_scores(golden_dir, kv_cache):
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run with -m 'not gpu' elsewhere"
    out, scores = run_generation_golden_case(golden_dir, "cuda", kv_cache)
    print("generated:", out[0, -8:].tolist(), "scores:", [round(float(s), 4) for s in scores])


def test_sparse_generation_reproduces_the_reference_tokens(golden_dir):
    """is_sparse = 2 on the gathered attention kernel (cogv_attention_fwd with kv_index): the 64 image codes the reference's
    sparse_attention_inference path generated, pivots drawn by `random.sample` under the same seed."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run with -m 'not gpu' elsewhere"
    out = run_sparse_generation_golden_case(golden_dir, "cuda")
    print("sparse generation:", out[0, -8:].tolist())
