"""Autoregressive filling of a token sequence with memories -- the reference's generation/sampling.py:25-209 on the
HIP model.  Host logic (filtering, sampling, beam bookkeeping) is torch; the model calls are the HIP path.  With
`GPT2Model(..., kv_cache=True)` the memories are per-layer key/value caches appended in place, so a step costs the new
positions only; with the reference's layer-input memories the same loop works unchanged (the K/V of the whole memory are
re-projected every step, as in mpu/sparse_transformer.py:135-140).

`seq`: 1-D tensor; ids >= 0 are given, -1 = generate one token, -N = generate with N beams.
`args`: object with .temperature, .top_k, .top_p, .is_sparse (0 or 2).
`tokenizer`: anything with the unified tokenizer's id-space interface (default: IdSpace(), the released layout)."""
import torch
import torch.nn.functional as F

from .id_space import IdSpace


def top_k_logits(logits, top_k=0, top_p=0.0, filter_value=-float('Inf')):
    """generation/sampling.py:25-50: keep the top_k largest logits and/or the smallest set whose probability mass
    exceeds top_p (the nucleus form works on one row, as in the reference)."""
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        row = logits.view(logits.size()[1]).contiguous()
        sorted_logits, sorted_indices = torch.sort(row, descending=True)
        cumulative = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative > top_p
        remove[..., 1:] = remove[..., :-1].clone()      # keep the first token that crosses the threshold
        remove[..., 0] = 0
        row[sorted_indices[remove]] = filter_value
        logits = row.view(1, -1).contiguous()
    return logits


def get_batch(context_tokens, device, args=None):
    """generation/sampling.py:52-63 with pretrain_gpt2.get_masks_and_position_ids (:265-299) for the plain case:
    left-to-right mask [1, 1, s, s] and positions 0..s-1."""
    tokens = context_tokens
    tokens = tokens.unsqueeze(0).contiguous() if tokens.dim() == 1 else tokens.view(tokens.shape[0], -1).contiguous()
    tokens = tokens.to(device)
    s = tokens.shape[1]
    attention_mask = torch.tril(torch.ones((1, s, s), device=device)).unsqueeze(1)
    position_ids = torch.arange(s, dtype=torch.long, device=device).unsqueeze(0).expand_as(tokens).clone()
    return tokens, attention_mask, position_ids


def shrink_beams(tokens, mems, nb, score):
    """generation/sampling.py:188-198: fall back to the best beam when the beam count changes."""
    if tokens.shape[0] == nb:
        return tokens, mems, score
    max_idx = score.index(max(score))
    return tokens[max_idx].unsqueeze(0), [mem[max_idx: max_idx + 1] for mem in mems], [0]


def add_interlacing_beam_marks(seq, nb=12, period=3000):
    """generation/sampling.py:200-211: turn runs of -1 into -nb, alternating the beam count every `period` tokens."""
    assert isinstance(seq, list) or len(seq.shape) == 1
    blk_cnt = 0
    for i in range(len(seq)):
        if seq[i] == -1:
            blk_cnt += 1
            seq[i] = -nb
            if blk_cnt == period:
                nb += (nb % 2) * 2 - 1
                blk_cnt = 0
        else:
            blk_cnt = 0


def filling_sequence(model, seq, args, mems=None, invalid_slices=[], tokenizer=None, **kwargs):
    """generation/sampling.py:65-186.  Returns the completed token rows [beams, len(seq)]."""
    tokenizer = tokenizer if tokenizer is not None else IdSpace()
    n_img, n_txt = tokenizer.img_tokenizer.num_tokens, tokenizer.txt_tokenizer.num_tokens
    boi, eoi = (tokenizer['[BOI1]'], tokenizer['[BOI2]']), (tokenizer['[EOI1]'], tokenizer['[EOI2]'])
    roi2 = tokenizer['[ROI2]']
    device = seq.device
    assert len(seq.shape) == 1
    out_seq_length = len(seq)
    seq_l = seq.tolist()                              # one host copy instead of a device sync per comparison
    context_length, offset = 0, 100000
    invalid_slices = [slice(0, n_img)]

    def slices_after(tok, current):
        if tok in boi:                                 # inside an image: only image codes may be generated
            return [slice(n_img, None)]
        if tok in eoi:                                 # after an image: only text pieces
            return [slice(0, n_img), slice(n_img + n_txt, None)]
        return current

    while seq_l[context_length] >= 0:
        invalid_slices = slices_after(seq_l[context_length], invalid_slices)
        if seq_l[context_length] == roi2:
            offset = context_length
        context_length += 1
    tokens, attention_mask, position_ids = get_batch(seq[:context_length], device, args)
    counter, index = context_length - 1, 0
    mems = [] if mems is None else mems
    score = [0]
    if args.is_sparse == 2:
        img_indices_bool = tokens < n_img
        txt_indices_bool = ~img_indices_bool
    elif args.is_sparse == 0:
        txt_indices_bool = img_indices_bool = None
    else:
        raise ValueError('set is_sparse==2 for inference.')

    with torch.no_grad():
        while counter < out_seq_length - 1:
            nxt = seq_l[counter + 1]
            invalid_slices = slices_after(nxt, invalid_slices)
            if index == 0:                                                     # the whole context at once
                position_ids[position_ids > offset] -= offset
                logits, *mems = model(tokens, position_ids, attention_mask, txt_indices_bool, img_indices_bool,
                                      args.is_sparse, *mems)
                index = counter
            elif nxt >= 0:                                                      # a given token: just append it
                if nxt == roi2:
                    offset = counter + 1
                tokens, mems, score = shrink_beams(tokens, mems, 1, score)
                counter += 1
                tokens = torch.cat((tokens, seq[counter: counter + 1].expand(tokens.shape[0], 1)), dim=1)
                if args.is_sparse == 2:
                    img_indices_bool = tokens < n_img
                    txt_indices_bool = ~img_indices_bool
                continue
            else:
                assert tokens.shape[1] == counter + 1
                position_ids = torch.arange(index, counter + 1, dtype=torch.long, device=device).unsqueeze(0)
                position_ids[position_ids > offset] -= offset
                tokens, mems, score = shrink_beams(tokens, mems, -nxt, score)
                logits, *mems = model(tokens[:, index:], position_ids, 0, txt_indices_bool, img_indices_bool,
                                      args.is_sparse, *mems)
                index = counter
            nb = -nxt
            counter += 1
            index += 1

            logits = logits[:, -1].float()                                      # [beams, vocab]
            logits /= args.temperature
            for invalid_slice in invalid_slices:
                logits[..., invalid_slice] = -float('Inf')
            logits = top_k_logits(logits, top_k=args.top_k, top_p=args.top_p)
            probs = F.softmax(logits, dim=-1)
            if nb > 1 and tokens.shape[0] == 1:                                 # 1 -> nb beams
                tokens = tokens.expand(nb, -1).contiguous()
                mems = [mem.expand(nb, -1, -1) for mem in mems]
                prev = torch.multinomial(probs, num_samples=nb, replacement=True)
                score = torch.log(torch.gather(probs, dim=1, index=prev)[0]).tolist()
            else:                                                               # nb -> nb
                assert tokens.shape[0] == nb
                prev = torch.multinomial(probs, num_samples=1)
                score_plus = torch.log(torch.gather(probs, dim=1, index=prev)[:, 0])
                for idx in range(nb):
                    score[idx] += score_plus[idx]
            tokens = torch.cat((tokens, prev.view(tokens.shape[0], 1)), dim=1)
            if args.is_sparse == 2:
                img_indices_bool = tokens < n_img
                txt_indices_bool = ~img_indices_bool
    return tokens.view(tokens.shape[0], -1).contiguous()


def inverse_prompt_score(model, seq, args, tokenizer=None):
    """generation/sampling.py:222-239 (post-selection): for rows laid out as [BASE] [BOI1] 1024 image codes [EOI1] [ROI1] text...,
    the log-likelihood of the text given the image -- one full forward per call, image codes excluded from the softmax,
    summed over the text positions.  Returns [rows] fp32."""
    tokenizer = tokenizer if tokenizer is not None else IdSpace()
    assert seq.dim() == 2
    first_text = 2 + 1024 + 1                                   # index of [ROI1]: the text it scores starts right after
    assert int(seq[0, first_text]) == tokenizer['[ROI1]']
    tokens, attention_mask, position_ids = get_batch(seq, seq.device, args)
    with torch.no_grad():
        logits, *_ = model(tokens, position_ids, attention_mask, None, None, args.is_sparse)
        logits = logits.float()
        logits[..., :tokenizer.img_tokenizer.num_tokens] = -float('Inf')
        log_probs = F.log_softmax(logits[:, first_text:-1], dim=-1)
        return torch.gather(log_probs, 2, tokens[:, first_text + 1:].unsqueeze(-1)).squeeze(-1).sum(dim=-1)


# (row block, column block, lines to fill) of the nine overlapping 16 x 16 -> 32 x 32 windows, in generation order
_MAGNIFY_WINDOWS = ((0, 0, 18), (0, 1, 30), (0, 2, 30), (1, 1, 30), (1, 0, 30), (1, 2, 30), (2, 0, 32), (2, 1, 32), (2, 2, 32))


def magnify(model, tokenizer, tokens_list, text_token_list, args, fill=None):
    """generation/magnify.py:22-43 (super-resolution): a 32 x 32 code map is magnified to 64 x 64 window by window -- each window
    conditions on the text, a 16 x 16 patch of the small map and the marker run [EOI1] [ROI2] [POS0] [BASE] [BOI2], and fills
    the not-yet-written lines of the matching 32 x 32 patch of the large map (lines written by an earlier window are given).
    Returns [1, 4096] codes.  `fill`: the sequence filler (default: filling_sequence of this module; the tokenizer is handed on)."""
    fill = fill if fill is not None else filling_sequence
    side = int(round(len(tokens_list) ** 0.5))
    assert side == 32 and side * side == len(tokens_list)
    code = tokens_list.view(side, side)
    midfix = torch.tensor([tokenizer[m] for m in ('[EOI1]', '[ROI2]', '[POS0]', '[BASE]', '[BOI2]')], device=code.device)
    big = torch.full((2 * side, 2 * side), -1, dtype=torch.long, device=code.device)
    only_image_codes = [slice(tokenizer.img_tokenizer.num_tokens, None)]
    for bi, bj, lines in _MAGNIFY_WINDOWS:
        patch = code[8 * bi: 8 * (bi + 2), 8 * bj: 8 * (bj + 2)].reshape(-1)
        target = big[16 * bi: 16 * bi + lines, 16 * bj: 16 * (bj + 2)]
        context = torch.cat([text_token_list, patch, midfix], dim=0)
        done = fill(model, torch.cat([context, target.reshape(-1)], dim=0), args, invalid_slices=only_image_codes, tokenizer=tokenizer)
        big[16 * bi: 16 * bi + lines, 16 * bj: 16 * (bj + 2)] = done[0, len(context):].view(lines, 32)
    return big.view(1, 4 * side * side)
