"""The reference's own driver functions executed over the mirrors (round-4 verdict, "boundary proven statically only").

tests/ref_drivers/drive_pretrain_gpt2.py imports /root/reference/pretrain_gpt2.py UNEDITED with the sys.modules aliases of
INTEGRATION.md section 2 and calls setup_model_and_optimizer / train_step (-> get_model, get_optimizer, get_batch,
forward_step, backward_step) four times on BASELINE configs[0]; the HIP entry points are replaced by tests/cpu_ops.py (no GPU
in the build container).  Runs in a subprocess so that the aliases never reach this session.  Skipped where /root/reference
is absent (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
needs_reference = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference only exists in the build container")


def _run(script, **env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_drivers", script)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    return json.loads(line[-1][7:])


@needs_reference
@pytest.mark.parametrize("checkpoint_activations", ["0", "1"])
def test_reference_train_step_runs_over_the_mirrors_and_reproduces_the_reference_loss(checkpoint_activations):
    """checkpoint_activations = 1: --checkpoint-activations as the reference's scripts set it (the mirror recomputes each layer inside
    its fused backward): the same numbers."""
    out = _run("drive_pretrain_gpt2.py", COGV_DRV_CHECKPOINT_ACTIVATIONS=checkpoint_activations)
    gold = out["golden"]                               # the reference's own fp32 modules on the same rows (gen_golden_cfg1.py)
    s1, s2, s3, s4 = out["step1"], out["step2"], out["step3"], out["step4"]
    # 1: default dynamic scale 2^32 overflows -> skipped, nothing moves, the scheduler does not advance, hysteresis 2 keeps the scale
    assert s1["skipped"] == 1 and s1["params_unchanged"] and s1["lr_steps"] == 0 and s1["scale_after"] == 2.0 ** 32
    assert abs(s1["loss"] - gold["loss"]) < 2e-3 * gold["loss"]
    # 2: real gradients: the loss and the global gradient norm of the reference itself; warm-up lr = 0 -> no movement
    assert s2["skipped"] == 0 and s2["lr_steps"] == 1 and s2["params_unchanged"] and s2["lr_next"] == 1.5e-4
    assert abs(s2["loss"] - gold["loss"]) < 2e-3 * gold["loss"]
    assert abs(s2["grad_norm"] - gold["grad_norm"]) < 5e-3 * gold["grad_norm"]
    assert s2["img_loss"] > 0 and s2["txt_loss"] > 0
    # 3 / 4: the first AdamW update with lr > 0 moves the table by ~lr and the loss drops on the same batch
    assert s3["skipped"] == 0 and 0.5e-4 < s3["max_param_change"] < 4e-4 and s3["adam_steps"] == 2
    assert s4["skipped"] == 0 and s4["loss"] < s3["loss"] - 0.05 and s4["lr_steps"] == 3
    # the reference's main loop (pretrain_gpt2.train): iterations 5..8 with logging every 2, checkpoints every 2 through the
    # reference's utils.save_checkpoint, validation (evaluate: eval mode, no_grad, back to train mode) at iteration 8
    t = out["train_loop"]
    assert t["iteration"] == 8 and t["lr_steps"] == 7 and t["adam_steps"] == 7
    assert t["saved"] == ["6", "8"] and t["tracker"] == "8"
    assert t["logged_iterations"] == [6, 8] and t["lm_losses"][1] < t["lm_losses"][0] < s4["loss"]
    assert len(t["validation"]) == 1 and t["training_mode_restored"]
    assert abs(t["validation"][0] - t["evaluate_again"]) < 1e-4 * t["evaluate_again"]      # same rows, no update in between
    assert t["evaluate_again"] < t["lm_losses"][1]                                          # evaluated after the last update


@needs_reference
def test_reference_generation_drivers_run_over_the_mirrors_and_reproduce_the_reference_tokens():
    """generate_samples.py (prepare_tokenizer, setup_model, generate_images_once, post_selection) and generation/sampling.py
    (filling_sequence, inverse_prompt_score), unedited, over the mirrors on the CPU-emulated ops: the 40 image codes both beams
    fill in are the ones the reference's own fp32 model produced under the same functions (oracle/gen_golden_generate.py ->
    tests/golden/generate_samples.npz), and the two post-selection scores agree to 5e-3 (sums of nine log-probabilities near
    -50).  A differing token is accepted only where the reference's two best admissible logits were closer than 0.004 of the
    logits' standard deviation -- a coin flip under 16-bit arithmetic -- and ends the comparison there."""
    import numpy as np
    out = _run("drive_generate_samples.py")
    gold = np.load(os.path.join(HERE, "golden", "generate_samples.npz"))
    want, gaps = gold["t2i_out"], gold["t2i_gaps"]
    n_gen = int((gold["t2i_seq"] < 0).sum())
    assert out["vocab"] == int(gold["vocab"])
    assert len(out["t2i_tokens"]) == want.shape[0] == 2                       # one DecodeIds call per beam
    for beam, ids in enumerate(out["t2i_tokens"]):
        assert len(ids) == want.shape[1]
        n_ctx = len(ids) - n_gen
        assert ids[:n_ctx] == want[beam, :n_ctx].tolist()
        for i in range(n_gen):
            if ids[n_ctx + i] != int(want[beam, n_ctx + i]):
                assert gaps[i] < 0.004, (beam, i, ids[n_ctx + i], int(want[beam, n_ctx + i]), float(gaps[i]))
                break
        else:
            assert ids == want[beam].tolist()
    assert float(gaps.min()) > 0.004                                          # i.e. with this fixture every token must match
    assert out["saved"] == ["0.jpg", "1.jpg", "concat.jpg"] and out["saved_concat_shape"][0] == 2
    assert out["sel_text"] == "two candidates"
    assert np.allclose(out["sel_scores"], gold["sel_scores"], rtol=0, atol=5e-3), (out["sel_scores"], gold["sel_scores"].tolist())
    assert out["sel_scores"][1] > out["sel_scores"][0]                        # the ranking post-selection exists for
    # sparse generation (is_sparse = 2): 64 codes past the 32-position trailing window, pivots drawn with random.sample per layer
    sp_want, sp_gaps = gold["sparse_out"], gold["sparse_gaps"]
    assert len(out["sparse_tokens"]) == 1 and float(sp_gaps.min()) > 0.004
    assert out["sparse_tokens"][0] == sp_want[0].tolist()
    # cogview_amd.generation's own inverse_prompt_score / magnify against the reference's functions: same model -> same scores;
    # a positional stand-in model -> the same 64 x 64 codes, token for token (window order, given lines, [ROI2] position offset)
    assert out["score_mirror_vs_reference_fn"] < 1e-4
    assert out["magnify_equal"] and out["magnify_shape"] == [1, 4096]


@needs_reference
def test_checkpoints_cross_the_boundary_both_ways_through_the_reference_utils(tmp_path):
    """utils.save_checkpoint / load_checkpoint of the reference, unedited (drive_checkpoint_interop.py, three processes):
    reference-written weights load into the mirrors (--finetune form) and give the reference's logits; a file written over the
    mirrors (weights + FP16_Optimizer state incl. the pickled loss scaler + AnnealingLR state) resumes the mirrors BIT-identically
    (step 4 of a run = step 4 after save / fresh model / load) and loads into the reference's own fp32 model, which then gives
    the mirrors' logits."""
    import numpy as np

    def run(mode):
        r = subprocess.run([sys.executable, os.path.join(HERE, "ref_drivers", "drive_checkpoint_interop.py"), mode, str(tmp_path)],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])

    def close(a, b, tol):
        ra, rb = np.array(a["rows"]), np.array(b["rows"])
        assert abs(a["norm"] - b["norm"]) < tol * b["norm"]
        assert np.linalg.norm(ra - rb) < tol * np.linalg.norm(rb), np.linalg.norm(ra - rb) / np.linalg.norm(rb)

    ref = run("ref_save")
    mir = run("mirror")
    back = run("ref_load")
    close(mir["logits_after_loading_reference_file"], ref["logits"], 2e-3)          # fp16 mirrors vs the fp32 reference
    close(mir["setup_model_deepspeed_layout"], ref["logits"], 2e-3)                   # generate_samples.setup_model, both branches
    close(mir["setup_model_plain"], ref["logits"], 2e-3)
    assert mir["losses"][2] < mir["losses"][0] - 0.3                                  # the three steps trained
    a, b = mir["step4_uninterrupted"], mir["step4_resumed"]
    assert a == b and mir["weights_after_step4_equal"], (a, b, mir["weights_after_step4_maxdiff"])
    assert a["adam_steps"] == 4 and a["lr_steps"] == 4
    assert mir["dropout_states_restored"]           # the rng block of the reference's save / load carries the mirrors' dropout states
    close(back["logits"], mir["logits_of_saved_model"], 2e-3)
    # and the file itself names the reference's classes only: it opens without this package on the path
    blob = open(os.path.join(str(tmp_path), "mirror", "3", "mp_rank_00_model_states.pt"), "rb").read()
    assert b"cogview_amd" not in blob and b"loss_scaler" in blob


def _run_ranks(script, world, port, mp, **env):
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), str(mp)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env)) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, so[-2000:] + "\n" + se[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return outs


@needs_reference
@pytest.mark.parametrize("fp32_allreduce", ["0", "1"])
def test_reference_train_step_with_its_own_data_parallel_wrapper(fp32_allreduce):
    """The reference's other setting, USE_TORCH_DDP = False (pretrain_gpt2.py:19, 38-41, 104-105, 371-375): model/distributed.py's
    own DistributedDataParallel, whose exchange backward_step requests with allreduce_params(reduce_after=False,
    fp32_allreduce=args.fp32_allreduce).  Two data-parallel ranks over the mirror of that class: same gradients (the golden's
    norm), same parameters, with the 16-bit and with the fp32 exchange."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = _run_ranks(os.path.join(HERE, "ref_drivers", "drive_pretrain_gpt2_dp2.py"), 2, port, 1, COGV_DRV_TORCH_DDP="0",
                      COGV_DRV_FP32_ALLREDUCE=fp32_allreduce)
    gold = outs[0]["golden"]
    for o in outs:
        s1, s2 = o["step1"], o["step2"]
        assert s1["skipped"] == 0 and s1["grads_equal_across_ranks"] and s1["params_equal_across_ranks"] and s2["params_equal_across_ranks"]
        assert abs(s1["loss_reduced"] - gold["loss"]) < 2e-3 * gold["loss"]
        assert abs(s1["grad_norm"] - gold["grad_norm"]) < 5e-3 * gold["grad_norm"]
    assert outs[0]["step1"] == outs[1]["step1"] and outs[0]["step2"] == outs[1]["step2"]


@needs_reference
def test_reference_train_loop_with_dropout_replays_its_masks_under_activation_checkpointing():
    """The reference's defaults -- hidden / attention dropout 0.1 -- through its train_step and its main loop over the mirrors (the
    CPU emulation draws the kernels' own counter-based masks, oracle/cogview_oracle.py dropout_keep_mask / attention_keep_mask):
    with --checkpoint-activations every layer is recomputed in backward and must see the masks of its forward pass, so every loss
    of the run is the same number with and without it; and dropout does act (other losses than the dropout-free run's)."""
    runs = [_run("drive_pretrain_gpt2.py", COGV_DRV_DROPOUT="0.1", COGV_DRV_CHECKPOINT_ACTIVATIONS=ck) for ck in ("0", "1")]
    seq = [[r["step2"]["loss"], r["step2"]["grad_norm"], r["step3"]["loss"], r["step4"]["loss"]] + r["train_loop"]["lm_losses"]
           + r["train_loop"]["validation"] for r in runs]
    assert seq[0] == seq[1], seq
    gold = runs[0]["golden"]
    assert abs(seq[0][0] - gold["loss"]) > 1e-3 and abs(seq[0][0] - gold["loss"]) < 0.05 * gold["loss"]
    assert seq[0][2] != seq[0][0]                                      # step 3 draws other masks than step 2 (the weights had not moved)
    assert runs[0]["step4"]["loss"] < runs[0]["step3"]["loss"] - 0.05


@needs_reference
@pytest.mark.parametrize("world,mp,dropout", [(2, 1, "0"), (2, 2, "0"), (4, 2, "0"), (4, 2, "0.1")])
def test_reference_train_step_on_several_ranks_over_the_mirrors(world, mp, dropout):
    """pretrain_gpt2.train_step, unedited, on two data-parallel ranks / one model split over two model-parallel ranks / both
    (four ranks) over the mirrors (one process per rank, gloo; the four golden rows shared out over the data-parallel ranks).
    The reference runs with USE_TORCH_DDP = True: its backward_step never calls allreduce_params, so the mirror's
    PyTorchDistributedDataParallel has to finish the exchange without being asked.  After the first step the data-parallel
    replicas hold the same gradients -- the mean over all four rows: the golden's global gradient norm, which the
    model-parallel ranks assemble from their shards (mpu/grads.py:28-74) -- and after every step the same parameters; the loss
    the script all-reduces is the golden's.  With dropout 0.1 (the reference's default; no golden then): the replicas still
    agree bit for bit, and so do the parameters every model-parallel rank holds a copy of (same hidden-dropout masks on the ranks of
    a model-parallel group, mpu/random.py:198-233)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(HERE, "ref_drivers", "drive_pretrain_gpt2_dp2.py")
    outs = _run_ranks(script, world, port, mp, COGV_DRV_DROPOUT=dropout)
    gold = outs[0]["golden"]
    for o in outs:
        s1, s2 = o["step1"], o["step2"]
        assert s1["skipped"] == 0 and s2["skipped"] == 0
        assert s1["grads_equal_across_ranks"] and s1["params_equal_across_ranks"] and s2["params_equal_across_ranks"]
        assert s2["replicated_params_equal_across_mp_ranks"]
        tol = 2e-3 if dropout == "0" else 2e-2
        assert abs(s1["loss_reduced"] - gold["loss"]) < tol * gold["loss"]
        if dropout == "0":
            assert abs(s1["grad_norm"] - gold["grad_norm"]) < 5e-3 * gold["grad_norm"]
        assert s2["loss_reduced"] < s1["loss_reduced"] - 0.05
    assert all(o["step1"] == outs[0]["step1"] and o["step2"] == outs[0]["step2"] for o in outs)
